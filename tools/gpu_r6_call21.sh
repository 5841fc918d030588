#!/bin/bash
# round 6, GPU call 21: the long determinism probes with the final round-6 code (conv_x3d, ASPP branch group, range planner)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call21; rm -rf $O; mkdir -p $O
(echo "tools/race_probe.py 3 40 (mfma_mode 3, batch-16 train step: one-stream reference, then 40 runs on the three-stream executor)"; timeout 900 python tools/race_probe.py 3 40 2>&1 | tail -12) > $O/race_probes.txt
(echo; echo "tools/race_probe_infer.py 200 (S30 inference, mode 3)"; timeout 600 python tools/race_probe_infer.py 200 2>&1 | tail -4) >> $O/race_probes.txt
(echo; echo "tools/race_probe_infer.py 100 1 (--tta)"; timeout 600 python tools/race_probe_infer.py 100 1 2>&1 | tail -4) >> $O/race_probes.txt
cat $O/race_probes.txt | tail -24
