#!/bin/bash
# round 5, GPU call 16: last sanity with the final tree -- kernel + train parity, smoke, the default bench line parsed from the tail
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call16; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench_stdout.txt 2> $O/bench_stderr.txt; echo "bench rc=$?"
python - <<'PY'
import json
t=open('gpurun_out/call16/bench_stdout.txt').read()
j=json.loads(t[-8000:].splitlines()[-1])
print('stdout bytes', len(t), 'value', j['value'], 'ms', j['ms_per_step'], 'frac', j['roofline']['frac'], 'traffic', j['roofline']['traffic'], 'tta', j['tta']['ms_per_step'], 'train', j['train']['ms_per_step'], j['train']['frac'])
PY
