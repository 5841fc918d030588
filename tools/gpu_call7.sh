#!/bin/bash
# round 5, GPU call 7: the long forms of the determinism probes with the new conv_x3h staging layout in the mix
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call7; rm -rf $O; mkdir -p $O
{ echo "tools/race_probe.py 3 40 (mfma_mode 3, batch-16 train step: one-stream reference, then 40 runs on the three-stream executor)";
  timeout 600 python tools/race_probe.py 3 40 2>&1 | grep -v amdgpu.ids | tail -4;
  echo; echo "tools/race_probe_infer.py 200 (S30 inference, mode 3)";
  timeout 600 python tools/race_probe_infer.py 200 2>&1 | grep -v amdgpu.ids | tail -3;
  echo; echo "tools/race_probe_infer.py 100 1 (--tta)";
  timeout 600 python tools/race_probe_infer.py 100 1 2>&1 | grep -v amdgpu.ids | tail -3; } > $O/race_probes.txt 2>&1
cat $O/race_probes.txt
