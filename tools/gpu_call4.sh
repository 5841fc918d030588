#!/bin/bash
# round 5, GPU call 4: fast uniform-chunk loader path of conv_x3h -- parity, bench (with the old path via VR_CONV_DBG=128, and with one more workgroup per CU)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call4; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-14s ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:3]))
PY
  cp gpurun_out/bench_detail.json $O/detail_$name.json
}
run infer_new infer VR_NOP=1
run infer_old infer VR_CONV_DBG=128
run infer_new2 infer VR_NOP=1
run infer_hi infer VR_X3H_HI=1
run train_new train VR_NOP=1
run train_old train VR_CONV_DBG=128
run train_hi train VR_X3H_HI=1
