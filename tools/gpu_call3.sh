#!/bin/bash
# round 5, GPU call 3: channel-innermost low-resolution staging in conv_x3h (fused upsample) -- parity, phase trace, bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call3; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_golden.py tests/test_gpu_train.py -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
VR_CONV_DBG=64 timeout 600 python tools/x3h_trace.py > $O/x3h_trace.txt 2>&1; echo "trace rc=$?"; grep -v Warning $O/x3h_trace.txt | head -60
for i in 1 2; do
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_infer$i.json 2> $O/bench_infer$i.err
python - "$O/bench_infer$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('infer ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:2]))
PY
done
cp gpurun_out/bench_detail.json $O/bench_detail_infer.json
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_train.json 2> $O/bench_train.err
python - "$O/bench_train.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('train ms_per_step %.3f  kernel_ms(serialised) %.3f  classes %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline']['classes'][:4]))
PY
cp gpurun_out/bench_detail.json $O/bench_detail_train.json
