#!/bin/bash
# round 5, GPU call 10: are the many-stream slowdowns (ASPP fan-out 12.4 vs 8.65 ms, three lanes 10.8) the HIP runtime's default of four hardware queues?
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call10; rm -rf $O; mkdir -p $O
run() { name=$1; mode=$2; shift; shift
  env "$@" timeout 300 python bench.py --mode $mode --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('%-18s ms_per_step %.3f  kernel_ms(serialised) %.3f' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step']))
PY
}
run nofan_q4 infer VR_ASPP_FAN=0
run nofan_q8 infer VR_ASPP_FAN=0 GPU_MAX_HW_QUEUES=8
run fan_q8 infer GPU_MAX_HW_QUEUES=8
run fan_q16 infer GPU_MAX_HW_QUEUES=16
run lanes3_q8 infer VR_ASPP_FAN=0 VR_LANES=3 GPU_MAX_HW_QUEUES=8
run nofan_q2 infer VR_ASPP_FAN=0 GPU_MAX_HW_QUEUES=2
run train_q4 train VR_ASPP_FAN=0
run train_q8 train VR_ASPP_FAN=0 GPU_MAX_HW_QUEUES=8
timeout 300 python -m pytest tests/test_gpu_hazard.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "hazard pytest rc=$?"; tail -2 $O/pytest.log
