#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j18; mkdir -p $O
for i in 1 2; do
for cfg in "0" "8"; do
  VR_CONV_DBG=$cfg timeout 300 python bench.py --mode infer --no-cpu-baseline > $O/infer_d$cfg.json 2> $O/infer_d$cfg.err
  python -c "import json;j=json.load(open('$O/infer_d$cfg.json'));print('infer dbg $cfg', j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"
done
done
for cfg in "0" "8"; do
  VR_CONV_DBG=$cfg timeout 300 python bench.py --mode train --no-cpu-baseline > $O/train_d$cfg.json 2> $O/train_d$cfg.err
  python -c "import json;j=json.load(open('$O/train_d$cfg.json'));print('train dbg $cfg', j['value'], j['ms_per_step'])"
done
