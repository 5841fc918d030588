#!/bin/bash
# round 6, GPU call 26: BatchNorm-backward reduce pass in the flat form of the apply pass (one 16-byte load of z and g per thread) -- parity + A / B (VR_BN_REDUCE_FLAT=0 = the row-looped kernel)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call26; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_b16.py tests/test_gpu_hazard.py tests/test_gpu_kernel_coverage.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
ks=[(k[0][:34], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'bn_' in k[0]]
print('%-8s ms_per_step %.3f  kernel_ms(serialised) %.3f %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
PY
}
run old VR_BN_REDUCE_FLAT=0
run flat VR_NOP=1
run old2 VR_BN_REDUCE_FLAT=0
run flat2 VR_NOP=1
