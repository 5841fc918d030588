#!/bin/bash
# round 6, GPU call 9: conv_x3d (16-column layers on the fp16 pipe, the four ASPP branches in one launch) -- parity, A / B on the S30
# inference step and on the train step
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call9; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3d.py -m gpu -q -p no:cacheprovider -s > $O/pytest_x3d.log 2>&1; echo "pytest x3d rc=$?"; grep -E "passed|failed|error" $O/pytest_x3d.log | tail -3; grep -E "^FAILED|^ERROR|Error|assert " $O/pytest_x3d.log | head -20
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][:60], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'x3d' in k[0] or 'conv_dma_kernel<3, 1' in k[0] or 'conv_dma_kernel<1, 1, 1, 1, 32, 8, 16' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run infer_off infer VR_CONV_X3D=0
run infer_x3d infer VR_NOP=1
run infer_single infer VR_ASPP_FUSED=0
run infer_off2 infer VR_CONV_X3D=0
run infer_x3d2 infer VR_NOP=1
run infer_mt32 infer VR_X3D_MT=32
run train_off train VR_CONV_X3D=0
run train_x3d train VR_NOP=1
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider -x > $O/pytest_rest.log 2>&1; echo "pytest rest rc=$?"; tail -5 $O/pytest_rest.log
