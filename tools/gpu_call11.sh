#!/bin/bash
# round 5, GPU call 11: dilated weight gradients with three per-kernel-row windows instead of one haloed window -- parity, train bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call11; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_b16.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train$i.json 2> $O/bench_train$i.err
python - "$O/bench_train$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('train ms_per_step %.3f  kernel_ms(serialised) %.3f classes %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:6]))
PY
done
python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_detail.json'))
for k in d['roofline']['kernels']:
    if 'wgrad_mfma' in k[0] or 'wgrad_ws' in k[0]: print(k)
PY
cp gpurun_out/bench_detail.json $O/detail_train.json
