#!/bin/bash
# round 5, GPU call 6: conv_x3h plain tilings without the (unused) staging tile in LDS; three workgroups per CU for 64x8 (VR_X3H_HI=1) re-measured
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call6; rm -rf $O; mkdir -p $O
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-14s ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:3]))
PY
  cp gpurun_out/bench_detail.json $O/detail_$name.json
}
run infer infer VR_NOP=1
run infer_hi infer VR_X3H_HI=1
run infer2 infer VR_NOP=1
run infer_hi2 infer VR_X3H_HI=1
run train train VR_NOP=1
run train_hi train VR_X3H_HI=1
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
