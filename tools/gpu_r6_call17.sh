#!/bin/bash
# round 6, GPU call 17: x3d tests after the pixel-offset change, the suite with VR_MFMA_MODE=0 for the two tests fixed since the evidence run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call17; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_golden.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
VR_MFMA_MODE=0 timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_kernel_coverage.py -m gpu -q -p no:cacheprovider > $O/pytest_m0.log 2>&1; echo "pytest mode0 rc=$?"; tail -3 $O/pytest_m0.log
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; j=json.loads(sys.stdin.read().splitlines()[-1]); print('infer', j['ms_per_step'])"
