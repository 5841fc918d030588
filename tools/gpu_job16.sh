#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j16; mkdir -p $O
echo "--- x6_check wino6 (MT by size)"; timeout 120 python tools/x6_check.py 2>&1 | grep -v amdgpu | cut -c1-60,100-240 | tail -8
echo "--- x6_check wino6 forced MT=64"; VR_WINO_MIN64=1 timeout 120 python tools/x6_check.py 2>&1 | grep -v amdgpu | cut -c1-60,100-240 | tail -8
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 VR_X6_VOL=1
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_$name.txt; echo "$name rc=$? $(grep -c vr-prof $O/pd_$name.txt)"
}
run m0 VR_MFMA_MODE=0
run w6 VR_MFMA_MODE=2
run w6_64 VR_MFMA_MODE=2 VR_X6_MIN_MT=64
run v2_64 VR_MFMA_MODE=2 VR_X6_MIN_MT=64 VR_WINO6=0 VR_X6_ILV=1
run w6_nomfma VR_MFMA_MODE=2 VR_CONV_DBG=2
run w6_notrans VR_MFMA_MODE=2 VR_CONV_DBG=3
run w6_nowdma VR_MFMA_MODE=2 VR_CONV_DBG=7
run w6_noepi VR_MFMA_MODE=2 VR_CONV_DBG=4
unset VR_NO_SIDE_STREAM VR_NO_SPLIT_BATCH VR_PROFILE_DUMP
for cfg in "0 32" "2 32" "2 64"; do
  set -- $cfg
  VR_MFMA_MODE=$1 VR_X6_MIN_MT=$2 timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer_m$1_$2.json 2> $O/infer_m$1_$2.err; echo "infer mode $1 minmt $2 rc=$?"
  python -c "import json;j=json.load(open('$O/infer_m$1_$2.json'));print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
VR_MFMA_MODE=2 timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x > $O/pytest_m2.log 2>&1; echo "pytest m2 rc=$?"; tail -4 $O/pytest_m2.log
