#!/bin/bash
# round 5, GPU call 2: where conv_x3h's time goes (phase stamps), host enqueue time, s_setprio / lane-count experiments, the two new tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call2; rm -rf $O; mkdir -p $O
VR_CONV_DBG=64 timeout 600 python tools/x3h_trace.py > $O/x3h_trace.txt 2>&1; echo "trace rc=$?"; cat $O/x3h_trace.txt
run() { # name, env...
  name=$1; shift
  env "$@" timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:3]))
PY
}
run base VR_NOP=1
run base2 VR_NOP=1
run prio VR_CONV_DBG=32
run lanes1 VR_LANES=1
run lanes3 VR_LANES=3
VR_ENQ_TIMING=1 timeout 300 python bench.py --mode infer --steps 10 --warmup 3 --no-cpu-baseline > /dev/null 2> $O/enq.err; grep "vr-enq" $O/enq.err | tail -12

timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -s -k "fp64_fixture or poisoned or witness" > $O/pytest_new.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed\|vs the fp64\|fp64 \|worst" $O/pytest_new.log | head -60
