#!/bin/bash
# round 6, GPU call 24: conv_dma -- one descriptor per uniform chunk, channels through the DMA's scalar offset (VR_CONV_DBG=8 = the per-channel form) -- parity + A / B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call24; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][16:50], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'conv_dma_kernel' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run infer_old infer VR_CONV_DBG=8
run infer_new infer VR_NOP=1
run infer_old2 infer VR_CONV_DBG=8
run infer_new2 infer VR_NOP=1
run train_old train VR_CONV_DBG=8
run train_new train VR_NOP=1
