#!/bin/bash
# round 6, GPU call 7: how the 11 crops of an S30 step are split over the two lanes (the host enqueues lane 0 first), and lane counts
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call7; rm -rf $O; mkdir -p $O
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode infer --steps 30 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-16s ms_per_step %.3f' % (sys.argv[2], j['ms_per_step']))
PY
}
run split_6_5 VR_LANE0_EXTRA=0
run split_7_4 VR_LANE0_EXTRA=1
run split_8_3 VR_LANE0_EXTRA=2
run split_6_5b VR_LANE0_EXTRA=0
run split_7_4b VR_LANE0_EXTRA=1
run lanes3 VR_LANES=3 VR_LANE0_EXTRA=0
run lanes3_e1 VR_LANES=3 VR_LANE0_EXTRA=1
run lanes1 VR_LANES=1
