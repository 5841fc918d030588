#!/bin/bash
# round-2 GPU job 1: full GPU test suite, the default bench line, train kernel trace + PMC passes, per-layer dumps
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j1; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -40 $O/pytest.log
timeout 600 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?"; tail -c 3000 $O/bench_all.json
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/train_dump.json 2> $O/train_dump.txt
VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode infer --steps 2 --warmup 1 --no-cpu-baseline > $O/infer_dump.json 2> $O/infer_dump.txt
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_train -o r -- python bench.py --mode train --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_train -o r -- python bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f_train.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_train -o r -- python bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w_train.log 2>&1
python tools/rocpd_summary.py $(ls $O/kt_train/*.db | head -1) $O/r02_train_kernel_trace_single_stream.md > /dev/null
python tools/pmc_summary.py $(ls $O/pmc_f_train/*.db | head -1) $(ls $O/pmc_w_train/*.db | head -1) 2 $O/r02_train_pmc.json train > $O/r02_train_pmc.md
find $O -name "*.db" -size +20M -delete
ls -la $O
