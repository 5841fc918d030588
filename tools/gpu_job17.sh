#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j17; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=5 -s > $O/pytest.log 2>&1; echo "pytest rc=$?"
grep -n "split mode\|mfma_mode\|passed\|failed\|error" $O/pytest.log | cut -c1-250 | tail -14
timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/j17/bench_all.json'))
print('infer', j['value'], j['ms_per_step'], j['roofline']['frac'], 'pcie', j['config'].get('pcie_inclusive_frames_per_sec'))
print('tta', j['tta']['value'], j['tta']['ms_per_step'])
print('train', j['train']['value'], j['train']['ms_per_step'], j['train']['roofline']['frac'])
print('train_bf16', j['train_bf16']['value'], j['train_bf16']['ms_per_step'])
print('split', j['split_bf16']['infer'], j['split_bf16']['train'])
PY
