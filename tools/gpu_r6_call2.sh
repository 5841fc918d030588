#!/bin/bash
# round 6, GPU call 2: SQ counters of conv_x3h vs conv_x3pp on three layers (harness mode 2), and the same layers on zero pixels (mode 4)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call2; rm -rf $O; mkdir -p $O
timeout 120 tools/_build/x3pp_proto 4 > $O/proto_zero.txt 2>&1; cut -c1-330 $O/proto_zero.txt
timeout 120 tools/_build/x3pp_proto 2 > $O/proto_m2.txt 2>&1; cut -c1-330 $O/proto_m2.txt
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
C2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"
C3="SQ_WAVE_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INST_CYCLES_VMEM SQ_INSTS_MFMA"
i=0
for C in "$C1" "$C2" "$C3"; do
  i=$((i+1))
  timeout 180 rocprofv3 --kernel-trace --pmc $C -d $O/p$i -o r -- tools/_build/x3pp_proto 2 > $O/p$i.log 2>&1
  python tools/pmc_dump.py $(ls $O/p$i/*.db | head -1) conv_x3 > $O/pmc$i.txt 2>> $O/p$i.log
done
find $O -name "*.db" -delete
cat $O/pmc1.txt | cut -c1-400
cat $O/pmc2.txt | cut -c1-400
cat $O/pmc3.txt | cut -c1-400
tail -3 $O/p3.log
