#!/bin/bash
# split-bf16 Winograd mode: correctness + first timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j11; mkdir -p $O
timeout 300 python tools/x6_check.py > $O/x6_check.log 2>&1; echo "x6_check rc=$?"; cat $O/x6_check.log | tail -12
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -p no:cacheprovider > $O/pytest_m0.log 2>&1; echo "pytest m0 rc=$?"; tail -3 $O/pytest_m0.log
VR_MFMA_MODE=2 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_train.py -q -p no:cacheprovider > $O/pytest_m2.log 2>&1; echo "pytest m2 rc=$?"; tail -8 $O/pytest_m2.log
for cfg in "0 32" "2 32" "2 64"; do
  set -- $cfg
  VR_MFMA_MODE=$1 VR_X6_MIN_MT=$2 timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer_m$1_$2.json 2> $O/infer_m$1_$2.err; echo "infer mode $1 minmt $2 rc=$?"
  python -c "import json;j=json.load(open('$O/infer_m$1_$2.json'));print(j['value'], j['ms_per_step'], j['roofline'])"
done
VR_MFMA_MODE=2 timeout 300 python bench.py --mode train --no-cpu-baseline > $O/train_m2.json 2> $O/train_m2.err; echo "train m2 rc=$?"
python -c "import json;j=json.load(open('$O/train_m2.json'));print(j['value'], j['ms_per_step'])"
VR_MFMA_MODE=2 VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_m2.txt
grep -c vr-prof $O/pd_m2.txt
