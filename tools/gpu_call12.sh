#!/bin/bash
# round 5, GPU call 12: HBM rates by read : write mix (tools/hbm_rw.hip); upsample backward with aligned 16-byte patch loads -- parity, A / B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call12; rm -rf $O; mkdir -p $O
tools/_build/hbm_rw > $O/hbm_rw.txt 2>&1; cat $O/hbm_rw.txt
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_hazard.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
up=[k for k in d['roofline']['kernels'] if 'upsample_bwd' in k[0]]
print('%-8s train ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], up))
PY
}
run vec VR_NOP=1
run novec VR_UPBWD_VEC=0
run vec2 VR_NOP=1
