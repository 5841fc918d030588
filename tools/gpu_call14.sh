#!/bin/bash
# round 5, GPU call 14: forward bilinear x2 through LDS (source rows fetched once per workgroup with aligned 16-byte loads) -- bit-equality with the row kernel, parity, A / B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call14; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_parity.py tests/test_gpu_train.py tests/test_golden.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
d=json.load(open('gpurun_out/bench_detail.json'))
up=[k for k in d['roofline']['kernels'] if 'upsample2x' in k[0]]
print('%-10s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], up))
PY
}
run t_lds train VR_NOP=1
run t_rows train VR_UP_LDS=0
run i_lds infer VR_NOP=1
run i_rows infer VR_UP_LDS=0
