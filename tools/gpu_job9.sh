#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j9; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log; grep -n "bf16 MFMA mode" $O/pytest.log | head -2
run() { timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline $1 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; }
for mb in 96 0 48 160; do echo -n "VR_BN_GROUP_MB=$mb  "; VR_BN_GROUP_MB=$mb run; done
