#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j19; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_train.py -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for i in 1 2; do
  timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer.json 2> $O/infer.err
  python -c "import json;j=json.load(open('$O/infer.json'));print('infer', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
timeout 300 python bench.py --mode train --no-cpu-baseline > $O/train.json 2> $O/train.err
python -c "import json;j=json.load(open('$O/train.json'));print('train', j['value'], j['ms_per_step'])"
VR_MFMA_MODE=2 timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer2.json 2> $O/infer2.err
python -c "import json;j=json.load(open('$O/infer2.json'));print('infer m2', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
