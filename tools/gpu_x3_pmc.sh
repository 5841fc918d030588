#!/bin/bash
# SQ / TCC counter passes over the stand-alone conv bench (tools/x3_proto.hip, mode 2 = the two full-resolution decoder layers)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-x3pmc}; mkdir -p $O
rocprofv3 -L 2>/dev/null | grep -o "SQ_[A-Z_0-9]*\|TCC_[A-Z_0-9a-z]*\|TCP_[A-Z_0-9a-z]*\|GRBM_[A-Z_]*" | sort -u > $O/counters.txt
C1="SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE"
C2="SQ_WAVE_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU"
C3="FETCH_SIZE"
C4="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
i=0
for C in "$C1" "$C2" "$C3" "$C4"; do
  i=$((i+1))
  timeout 120 rocprofv3 --kernel-trace --pmc $C -d $O/p$i -o r -- tools/x3_proto.bin 2 > $O/p$i.log 2>&1
  python tools/pmc_dump.py $(ls $O/p$i/*.db | head -1) > $O/pmc$i.txt 2>> $O/p$i.log
done
find $O -name "*.db" -delete
cat $O/pmc1.txt | cut -c1-330
