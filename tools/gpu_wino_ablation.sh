#!/bin/bash
# Where the Winograd conv kernel spends its time (profiles/README.md, round 2): per-layer HIP-event times of one single-stream
# inference step with parts of conv_wino_kernel switched off (VR_CONV_DBG: 2 no MFMA, 3 no input transform, 4 no epilogue,
# 6 no DMA, 7 no weight DMA), for the fp32-MFMA mode and the split-bf16 mode.  Results are wrong by construction; only times count.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/wino_ablation; mkdir -p $O
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1 VR_PROFILE_DUMP=1
for mode in 0 2; do
  for dbg in 0 2 3 4 6 7; do
    VR_MFMA_MODE=$mode VR_CONV_DBG=$dbg timeout 200 python bench.py --mode infer --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pd_m${mode}_d${dbg}.txt
    echo "mode $mode dbg $dbg: $(grep -c vr-prof $O/pd_m${mode}_d${dbg}.txt) launches, $(grep vr-prof $O/pd_m${mode}_d${dbg}.txt | awk '{s+=$(NF-5)} END {print s}') us"
  done
done
