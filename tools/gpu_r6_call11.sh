#!/bin/bash
# round 6, GPU call 11: per-launch table of one profiled S30 inference step and one train step (VR_PROFILE_DUMP), x3d test after the bar fix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call11; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_x3d.py -m gpu -q -p no:cacheprovider -s -k train_step > $O/pytest_x3d.log 2>&1; echo "pytest x3d rc=$?"; grep -E "passed|failed|largest|loss:" $O/pytest_x3d.log | tail -4
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode infer --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_infer.json 2> $O/dump_infer.txt; echo "infer rc=$?"
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_train.json 2> $O/dump_train.txt; echo "train rc=$?"
grep -c "vr-prof" $O/dump_infer.txt $O/dump_train.txt
