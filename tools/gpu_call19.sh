#!/bin/bash
# round 5, GPU call 19: materialize4 / bn_bwd_apply4 with the plane on blockIdx.y (one integer division per thread instead of three) -- parity, A / B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call19; rm -rf $O; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_golden.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
up=[k[:3] for k in d['roofline']['kernels'] if 'materialize' in k[0] or 'bn_bwd_apply' in k[0]]
print('%-8s train ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], up))
PY
}
run planes VR_NOP=1
run flat VR_MAT_PLANES=0
