#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j20; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -q -p no:cacheprovider -x -k "winograd or full_net or taps" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer.json 2> $O/infer.err
  python -c "import json;j=json.load(open('$O/infer.json'));print('infer', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
timeout 300 python bench.py --mode train --no-cpu-baseline > $O/train.json 2> $O/train.err
python -c "import json;j=json.load(open('$O/train.json'));print('train', j['value'], j['ms_per_step'])"
VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1 timeout 200 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_m0.txt
