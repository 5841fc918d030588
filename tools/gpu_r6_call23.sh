#!/bin/bash
# round 6, GPU call 23: conv_dma 1x1 (one 32 x 32 tile per wave) with the odd k-steps in a second accumulator -- is the class bound by its single dependent chain of matrix instructions?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call23; rm -rf $O; mkdir -p $O
VR_DMA_SPLITK=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "conv" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][:52], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'conv_dma_kernel<1' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run infer_old infer VR_NOP=1
run infer_sk infer VR_DMA_SPLITK=1
run infer_old2 infer VR_NOP=1
run infer_sk2 infer VR_DMA_SPLITK=1
run train_old train VR_NOP=1
run train_sk train VR_DMA_SPLITK=1
