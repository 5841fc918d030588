cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/sqx; mkdir -p $O
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --pmc $C -d $O/sq_cal -o r -- /tmp/mfma_peak > $O/sq_cal.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/sq_train -o r -- python bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline > $O/sq_train.log 2>&1
python tools/pmc_sq_summary.py $(ls $O/sq_train/*.db | head -1) $(ls $O/sq_cal/*.db | head -1) $O/train_sq_pmc.json > $O/train_sq_pmc.md
C2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA"
timeout 300 rocprofv3 --kernel-trace --pmc $C2 -d $O/sq2_train -o r -- python bench.py --mode train --steps 1 --warmup 0 --no-cpu-baseline > $O/sq2_train.log 2>&1
python tools/pmc_dump.py $(ls $O/sq2_train/*.db | head -1) wgrad_x3h > $O/sq2_wxh.txt 2>&1
python tools/pmc_dump.py $(ls $O/sq2_train/*.db | head -1) conv_x3h > $O/sq2_x3h.txt 2>&1
find $O -name "*.db" -delete; rm -rf $O/sq_cal $O/sq_train $O/sq2_train
