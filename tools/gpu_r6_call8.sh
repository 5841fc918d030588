#!/bin/bash
# round 6, GPU call 8: deferred + batched weight-gradient slab sums -- train parity tests, A / B on the batch-16 train step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call8; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_train.py tests/test_gpu_configs.py tests/test_gpu_b16.py tests/test_gpu_frontend.py -m gpu -q -p no:cacheprovider -x -k "train or b16 or frontend or step" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
ks=[k[:3] for k in d['roofline']['kernels'] if 'wgrad_reduce' in k[0]]
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f launches %s %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline'].get('launches_per_step'), ks))
PY
}
run immediate VR_NO_WGRAD_DEFER=1
run deferred VR_NOP=1
run immediate2 VR_NO_WGRAD_DEFER=1
run deferred2 VR_NOP=1
