#!/bin/bash
# round 5, first GPU call: the suite after the clean-up, the default bench line (must parse from the last 8000 characters), the fp64 batch-16 fixture
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call1; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 -s > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
grep -n "passed\|failed\|error" $O/pytest.log | tail -4
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"
python - <<'PY'
import json
t=open('gpurun_out/call1/bench_stdout.txt').read()
print('stdout bytes', len(t))
j=json.loads(t[-8000:].splitlines()[-1])
print('value', j['value'], 'ms', j['ms_per_step'], 'roof', j['roofline']['frac'], j['roofline']['kernel'], 'tta', j['tta']['ms_per_step'], 'train', j['train']['ms_per_step'], j['train']['frac'])
print(j['roofline']['classes']); print(j['train']['classes']); print(j['cpu_baseline'])
PY
cp gpurun_out/bench_detail.json $O/bench_detail.json
timeout 600 python tests/golden/make_golden_b16.py gpurun_out/call1/b16_fp64_small_grads.npz > $O/golden_b16.log 2>&1; echo "golden rc=$?"; tail -25 $O/golden_b16.log
