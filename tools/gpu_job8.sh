#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j8; mkdir -p $O
python tools/ww_debug.py 2>&1 | grep -v "^  " | tail -8
timeout 900 python -m pytest tests/test_gpu_train.py "tests/test_gpu_configs.py::test_full_net_train_step" -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/pytest.log
run() { timeout 300 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline $1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['achieved'])"; }
echo -n "WS default         "; run
echo -n "VR_WW_WS=0         "; VR_WW_WS=0 run
echo -n "WS bf16            "; run --bf16
echo -n "VR_WW_WS=0 bf16    "; VR_WW_WS=0 run --bf16
echo -n "WS DBG=2 (no MFMA) "; VR_WW_DBG=2 run
