#!/bin/bash
# round 6, GPU call 25: third workgroup per CU for the plain 64 x 8 conv_x3h tiling (VR_X3H_HI=1) and a 7 + 4 lane split, re-measured on the final code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call25; rm -rf $O; mkdir -p $O
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step']))
PY
}
run base infer VR_NOP=1
run hi infer VR_X3H_HI=1
run lane7 infer VR_LANE0_EXTRA=1
run base2 infer VR_NOP=1
run hi2 infer VR_X3H_HI=1
run lane7b infer VR_LANE0_EXTRA=1
