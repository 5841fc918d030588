#!/bin/bash
# round 6, GPU call 14: stride-2 weight gradient on wgrad_mfma (LDS-DMA loader) instead of wgrad_ws -- A / B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call14; rm -rf $O; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][:44], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'wgrad_ws' in k[0] or 'wgrad_mfma_kernel<3, 2' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base VR_NOP=1
run nows VR_WGRAD_WS=0
run base2 VR_NOP=1
