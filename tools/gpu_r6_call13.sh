#!/bin/bash
# round 6, GPU call 13: row-batched guarded epilogue, ASPP branch group + single materialisation in training, Winograd weight-gradient slab count knob
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call13; rm -rf $O; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_train.py tests/test_gpu_b16.py tests/test_gpu_configs.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -s > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|conv_x3d vs off|control|loss:" $O/pytest.log | tail -8; grep -E "^FAILED|^ERROR" $O/pytest.log | head
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][:44], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'wgrad_wino' in k[0] or 'reduce' in k[0] or 'x3d' in k[0] or 'conv_x3h_kernel<32, 16' in k[0] or 'materialize' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base VR_NOP=1
run p256 VR_WW_PTARGET=256
run p384 VR_WW_PTARGET=384
run nogroup VR_ASPP_FUSED=0
run base2 VR_NOP=1
