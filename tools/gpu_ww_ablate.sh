#!/bin/bash
# wgrad_wino_kernel phase ablations (VR_WW_DBG bits: 1 no transform, 2 no MFMA, 8 / 16 no input / dz transform): total time of the
# three instantiations over one serialised train step (rocprofv3 kernel trace).  Results are perf-only (the numbers are wrong).
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
O=gpurun_out/ww; mkdir -p $O
for d in 0 1 2 3 8 16; do
  VR_WW_DBG=$d timeout 300 rocprofv3 --kernel-trace --stats -d $O/kt_$d -o r -- python bench.py --mode train --steps 1 --warmup 1 --no-cpu-baseline > $O/kt_$d.log 2>&1
  python tools/rocpd_summary.py $(ls $O/kt_$d/*.db | head -1) $O/trace_$d.md > /dev/null
  echo "VR_WW_DBG=$d: $(grep wgrad_wino $O/trace_$d.md | awk -F'|' '{c+=$3; t+=$4} END {printf "%d launches %.2f ms", c, t}')"
done
find $O -name "*.db" -delete
