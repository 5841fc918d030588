#!/bin/bash
# round 6, GPU call 20: pixel-range count from the dispatch geometry also for the stride-2 (wgrad_ws) and dilated (wgrad_mfma) weight gradients; A / B against 512 workgroups
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call20; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for pt in 512 0 512 0; do
VR_WG_PTARGET=$pt timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$pt.json 2> $O/bench_$pt.err
python - $O/bench_$pt.json $pt <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
ks=[(k[0][:40], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'wgrad' in k[0]][:9]
print('WG_PTARGET', sys.argv[2], 'step %.2f ms' % j['ms_per_step'], 'serial %.2f' % j['roofline']['kernel_ms_per_step'], ks)
PY
done
