#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j7; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -6 $O/pytest.log
run() { timeout 300 python bench.py --mode $1 --steps 15 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
echo -n "infer default      "; run infer
echo -n "infer NO_UP_ROWS   "; VR_NO_UP_ROWS=1 run infer
echo -n "train default      "; run train
echo -n "train NO_UP_ROWS   "; VR_NO_UP_ROWS=1 run train
