#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_kernels.py "tests/test_gpu_configs.py::test_full_net_train_step" -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -25 $O/pytest.log
for v in 0 1; do
  echo "VR_WGRAD_WINO=$v"
  VR_WGRAD_WINO=$v timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline 2> $O/train_$v.err | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['achieved'])"
done
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/train_dump.json 2> $O/train_dump.txt
for d in 1 2 4; do echo "VR_WW_DBG=$d"; VR_WW_DBG=$d timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; done
