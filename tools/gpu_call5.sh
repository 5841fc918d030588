#!/bin/bash
# round 5, GPU call 5: BatchNorm backward with the finalize folded into the apply pass -- kernel + train parity, train bench (fused / unfused)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_b16.py tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-14s ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:4]))
PY
  cp gpurun_out/bench_detail.json $O/detail_$name.json
}
run train_fused train VR_NOP=1
run train_unfused train VR_BN_BWD_FUSED=0
run train_fused2 train VR_NOP=1
