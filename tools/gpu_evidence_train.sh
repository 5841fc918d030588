#!/bin/bash
# Partial refresh of the round's evidence after a train-only kernel change: default bench line + the train-step trace / counter passes.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/evidence; mkdir -p $O
timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?"
python - <<'PY'
import json
j=json.load(open('gpurun_out/evidence/bench_all.json'))
print('infer', j['value'], j['ms_per_step'], j['roofline']['frac'], 'pcie', j['config'].get('pcie_inclusive_frames_per_sec'))
print('tta', j['tta']['value'], j['tta']['ms_per_step'], j['tta']['roofline']['frac'])
print('train', j['train']['value'], j['train']['ms_per_step'], j['train']['roofline']['frac'])
print('train_bf16', j['train_bf16']['value'], j['train_bf16']['ms_per_step'])
print('fp32_mfma', j['fp32_mfma']['infer'], j['fp32_mfma']['train'])
print('cpu', j['cpu_baseline']['value'], j['train']['cpu_baseline']['value'])
PY
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
m=train
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_$m -o r -- python bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_$m.log 2>&1
python tools/rocpd_summary.py $(ls $O/kt_$m/*.db | head -1) $O/${m}_kernel_trace.md > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f_$m.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w_$m.log 2>&1
python tools/pmc_summary.py $(ls $O/pmc_f_$m/*.db | head -1) $(ls $O/pmc_w_$m/*.db | head -1) 2 $O/${m}_pmc.json $m > $O/${m}_pmc.md
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/sq_cal -o r -- /tmp/mfma_peak > $O/sq_cal.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/sq_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/sq_$m.log 2>&1
python tools/pmc_sq_summary.py $(ls $O/sq_$m/*.db | head -1) $(ls $O/sq_cal/*.db | head -1) $O/${m}_sq_pmc.json > $O/${m}_sq_pmc.md
find $O -name "*.db" -delete
head -14 $O/train_kernel_trace.md; tail -1 $O/train_kernel_trace.md; head -3 $O/train_pmc.md; grep "all kernels" $O/train_sq_pmc.md
