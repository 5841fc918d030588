#!/bin/bash
# round 5, GPU call 8: LSTM gate-bias sums cached in eval -- inference parity + bench; the train determinism probe again with its whole output
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call8; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_frontend.py tests/test_dropin_launcher.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2 3; do
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_infer$i.json 2> $O/bench_infer$i.err
python - "$O/bench_infer$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('infer ms_per_step %.3f  kernel_ms(serialised) %.3f launches %d classes %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['launches_per_step'], j['roofline']['classes'][:2]))
PY
done
timeout 600 python tools/race_probe.py 3 40 2>&1 | grep -v amdgpu.ids > $O/race_probe_train.txt; grep -c "0 tensors differ" $O/race_probe_train.txt; grep -v "0 tensors differ" $O/race_probe_train.txt | head
