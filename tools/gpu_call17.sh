#!/bin/bash
# round 5, GPU call 17: upsample backward with the interpolation weights shared through LDS -- parity, train bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call17; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_hazard.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
for i in 1 2; do
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_train$i.json 2> $O/bench_train$i.err
python - "$O/bench_train$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
up=[k[:3] for k in d['roofline']['kernels'] if 'upsample' in k[0]]
print('train ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], up))
PY
done
