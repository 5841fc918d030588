#!/bin/bash
# round 5, GPU call 15: BatchNorm-backward reduction with smaller / more chunks (VR_BN_CHUNK, VR_BN_MAXCH)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call15; rm -rf $O; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
up=[k[:3] for k in d['roofline']['kernels'] if 'bn_bwd' in k[0]]
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], up))
PY
}
run base VR_NOP=1
run c4096 VR_BN_CHUNK=4096 VR_BN_MAXCH=1024
run c8192 VR_BN_CHUNK=8192 VR_BN_MAXCH=512
run c32768 VR_BN_CHUNK=32768 VR_BN_MAXCH=256
run base2 VR_NOP=1
