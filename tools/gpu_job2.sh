#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py "tests/test_gpu_configs.py::test_full_net_train_step" -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -8 $O/pytest.log
run() { timeout 300 python bench.py --mode train --steps 8 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['achieved'])"; }
echo -n "default            "; run
echo -n "VR_WGRAD_GEMM=0    "; VR_WGRAD_GEMM=0 run
echo -n "VR_NO_UPBWD_TILED  "; VR_NO_UPBWD_TILED=1 run
echo -n "VR_WGRAD_WINO=0    "; VR_WGRAD_WINO=0 run
