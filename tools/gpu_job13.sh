#!/bin/bash
# where does the split-mode Winograd kernel spend its time: ablations (single stream, per-layer dump)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j13; mkdir -p $O
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 VR_X6_VOL=1 VR_X6_MIN_MT=64
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_$name.txt; echo "$name rc=$? $(grep -c vr-prof $O/pd_$name.txt)"
}
run m0 VR_MFMA_MODE=0
run m0_rot VR_MFMA_MODE=0 VR_WINO_ROT=1
run m2 VR_MFMA_MODE=2
run m2_rot VR_MFMA_MODE=2 VR_WINO_ROT=1
run m2_nowdma VR_MFMA_MODE=2 VR_CONV_DBG=7
run m2_nodma VR_MFMA_MODE=2 VR_CONV_DBG=6
run m2_nomfma VR_MFMA_MODE=2 VR_CONV_DBG=2
run m2_notrans VR_MFMA_MODE=2 VR_CONV_DBG=3
run m0_nowdma VR_MFMA_MODE=0 VR_CONV_DBG=7
run m0_nomfma VR_MFMA_MODE=0 VR_CONV_DBG=2
