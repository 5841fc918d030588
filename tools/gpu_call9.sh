#!/bin/bash
# round 5, GPU call 9: ASPP branch convs on four streams per lane (VR_ASPP_FAN) -- bit-equality / parity tests, bench with and without
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call9; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_hazard.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step']))
PY
}
run fan infer VR_NOP=1
run nofan infer VR_ASPP_FAN=0
run fan2 infer VR_NOP=1
run nofan2 infer VR_ASPP_FAN=0
run tta_fan tta VR_NOP=1
run tta_nofan tta VR_ASPP_FAN=0
timeout 300 python tools/race_probe_infer.py 100 2>&1 | grep -v amdgpu.ids | tail -2
