#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j14; mkdir -p $O
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 VR_X6_VOL=1 VR_X6_MIN_MT=64
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_$name.txt; echo "$name rc=$? $(grep -c vr-prof $O/pd_$name.txt)"
}
run m0 VR_MFMA_MODE=0
run m0_noepi VR_MFMA_MODE=0 VR_CONV_DBG=4
run m2 VR_MFMA_MODE=2
run m2_vol0 VR_MFMA_MODE=2 VR_X6_VOL=0
run m2_noepi VR_MFMA_MODE=2 VR_CONV_DBG=4
run m2_nodma VR_MFMA_MODE=2 VR_CONV_DBG=6
run m2_nomfma VR_MFMA_MODE=2 VR_CONV_DBG=2
run m2_notrans VR_MFMA_MODE=2 VR_CONV_DBG=3
run m2_mt32 VR_MFMA_MODE=2 VR_X6_MIN_MT=32
timeout 120 python tools/x6_check.py 2>&1 | grep -v amdgpu | cut -c1-60,100-240 | tail -3
