#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j4; mkdir -p $O
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_train -o r -- python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_train.log 2>&1
python tools/rocpd_summary.py $(ls $O/kt_train/*.db | head -1) $O/train_trace.md > /dev/null
find $O -name "*.db" -size +20M -delete
head -45 $O/train_trace.md
