#!/bin/bash
# round 6, GPU call 6: quad-form BiLSTM forward -- parity (kernel hook, taps, full net, train step), A / B on the S30 inference step and the train step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call6; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -x -k "lstm or taps or full_net or train_step or small_net" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
ks=[k[:3] for k in d['roofline']['kernels'] if 'lstm' in k[0]]
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
PY
}
run infer_reg infer VR_LSTM_QUAD=0
run infer_quad infer VR_NOP=1
run infer_reg2 infer VR_LSTM_QUAD=0
run infer_quad2 infer VR_NOP=1
run train_reg train VR_LSTM_QUAD=0
run train_quad train VR_NOP=1
