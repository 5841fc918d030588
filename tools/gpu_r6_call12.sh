#!/bin/bash
# round 6, GPU call 12: vectorised batched slab sum; small-Cin / small-Cout layers off the Winograd weight gradient (knobs); x3d train A/B test
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call12; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3d.py tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider -s -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed|largest|loss:|  stg" $O/pytest.log | tail -12
run() { name=$1; shift
  env "$@" timeout 300 python bench.py --mode train --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_$name.json 2> $O/bench_$name.err
  python - "$O/bench_$name.json" "$name" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().splitlines()[-1])
    d=json.load(open('gpurun_out/bench_detail.json'))
    ks=[(k[0][:44], k[1], round(k[2],3)) for k in d['roofline']['kernels'] if 'wgrad' in k[0]]
    print('%-12s ms_per_step %.3f  kernel_ms(serialised) %.3f  %s' % (sys.argv[2], j['ms_per_step'], j['roofline']['kernel_ms_per_step'], ks))
except Exception as e:
    print(sys.argv[2], 'FAILED', e)
PY
}
run base VR_NOP=1
run mincin4 VR_WW_MIN_CIN=4
run mincin16 VR_WW_MIN_CIN=16
run mincout32 VR_WW_MIN_COUT=32
run base2 VR_NOP=1
