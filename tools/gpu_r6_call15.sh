#!/bin/bash
# round 6, GPU call 15: bn_bwd_finalize folded into the last reduce block of a channel, stride-2 class weights in one launch -- train tests + bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call15; rm -rf $O; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_kernels.py tests/test_gpu_train.py tests/test_gpu_b16.py tests/test_gpu_configs.py tests/test_gpu_hazard.py tests/test_gpu_frontend.py tests/test_gpu_kernel_coverage.py tests/test_gpu_x3d.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.log
for i in 1 2 3; do
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$i.json 2> $O/bench_$i.err
python - "$O/bench_$i.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('train ms_per_step %.3f  kernel_ms(serialised) %.3f launches %s' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step'], j['roofline'].get('launches_per_step')))
PY
done
timeout 300 python bench.py --mode infer --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_infer.json 2> $O/bench_infer.err
python - "$O/bench_infer.json" <<'PY'
import json,sys
j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
print('infer ms_per_step %.3f  kernel_ms(serialised) %.3f' % (j['ms_per_step'], j['roofline']['kernel_ms_per_step']))
PY
