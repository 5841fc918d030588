#!/bin/bash
# round 6, last call: per-launch tables of one profiled step (final code), then the evidence run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6final; rm -rf $O; mkdir -p $O
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode infer --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_infer.json 2> $O/dump_infer.txt; echo "infer rc=$?"
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/dump_train.txt; echo "train rc=$?"
bash tools/gpu_evidence_r06.sh
