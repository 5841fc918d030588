#!/bin/bash
# split-bf16 Winograd mode v2 (transform-time split): lane-swap semantics, correctness, timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j12; mkdir -p $O
hipcc --offload-arch=gfx950 -O2 tools/permlane_test.hip -o /tmp/pl 2>/dev/null && /tmp/pl | tee $O/permlane.log
for v in 0 1; do
  VR_X6_VOL=$v timeout 300 python tools/x6_check.py > $O/x6_check_v$v.log 2>&1; echo "x6_check vol=$v rc=$?"; grep -v amdgpu.ids $O/x6_check_v$v.log | cut -c1-60,100-240 | tail -8
done
for cfg in "0 32 0" "2 64 0" "2 64 1" "2 32 0" "2 32 1"; do
  set -- $cfg
  VR_MFMA_MODE=$1 VR_X6_MIN_MT=$2 VR_X6_VOL=$3 timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer_m$1_$2_$3.json 2> $O/infer_m$1_$2_$3.err; echo "infer mode $1 minmt $2 vol $3 rc=$?"
  python -c "import json;j=json.load(open('$O/infer_m$1_$2_$3.json'));print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
VR_MFMA_MODE=2 VR_X6_VOL=1 VR_X6_MIN_MT=64 timeout 300 python bench.py --mode train --no-cpu-baseline > $O/train_m2.json 2> $O/train_m2.err; echo "train m2 rc=$?"
python -c "import json;j=json.load(open('$O/train_m2.json'));print(j['value'], j['ms_per_step'])"
VR_MFMA_MODE=2 VR_X6_VOL=1 VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_m2.txt
VR_MFMA_MODE=2 VR_X6_VOL=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -q -p no:cacheprovider -x > $O/pytest_m2.log 2>&1; echo "pytest m2 rc=$?"; tail -4 $O/pytest_m2.log
