#!/bin/bash
# Round-6 evidence on one MI355X (gpurun -- 'bash tools/gpu_evidence_r06.sh'): the full GPU suite (default mode and VR_MFMA_MODE=0), the
# default bench line, single-stream kernel traces + HBM PMC passes (inference, --tta, train step), the SQ counter passes, and the
# CONCURRENT timelines of one inference call and one train step (tools/timeline.py).  Everything lands under gpurun_out/evidence.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/evidence; rm -rf $O; mkdir -p $O
# (SKIP_TESTS=1: profiles only -- the two suite runs take 5 of the 9 minutes)
if [ -z "$SKIP_TESTS" ]; then
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 -s > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
grep -n "passed\|failed\|error\|rc=" $O/pytest.log | tail -4
fi
timeout 900 python bench.py > $O/bench_all.json 2> $O/bench_all.err; echo "bench rc=$?"
python - <<'PY'
import json
t=open('gpurun_out/evidence/bench_all.json').read()
print('stdout bytes', len(t))
j=json.loads(t[-8000:].splitlines()[-1])
print('infer', j['value'], j['ms_per_step'], j['roofline']['frac'], j['roofline']['frac_fp32_equivalent'], 'pcie', j['config'].get('pcie_inclusive_frames_per_sec'))
print('tta', j['tta']['value'], j['tta']['ms_per_step'], j['tta']['frac'])
print('train', j['train']['value'], j['train']['ms_per_step'], j['train']['frac'], j['train']['kernel'])
print('train_bf16', j['train_bf16']['value'], j['train_bf16']['ms_per_step'])
print('fp32_mfma', j['fp32_mfma']['infer'], j['fp32_mfma']['train'])
print('cpu', j['cpu_baseline']['value'], j['train']['cpu_baseline']['value'])
PY
cp gpurun_out/bench_detail.json $O/bench_detail.json
# concurrent timelines (the executor as it is timed): one inference call, one train step
timeout 300 rocprofv3 --kernel-trace -d $O/ktc_infer -o r -- python bench.py --mode infer --steps 6 --warmup 2 --no-cpu-baseline > $O/ktc_infer.log 2>&1
python tools/timeline.py $(ls $O/ktc_infer/*.db | head -1) 5 vr::stft_tile start > $O/infer_timeline_concurrent.txt 2>&1
timeout 300 rocprofv3 --kernel-trace -d $O/ktc_train -o r -- python bench.py --mode train --steps 4 --warmup 2 --no-cpu-baseline > $O/ktc_train.log 2>&1
python tools/timeline.py $(ls $O/ktc_train/*.db | head -1) 5 > $O/train_timeline_concurrent.txt 2>&1
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
for m in train infer tta; do
  timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_$m -o r -- python bench.py --mode $m --steps 2 --warmup 1 --no-cpu-baseline > $O/kt_$m.log 2>&1
  python tools/rocpd_summary.py $(ls $O/kt_$m/*.db | head -1) $O/${m}_kernel_trace.md > /dev/null
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f_$m.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w_$m.log 2>&1
  python tools/pmc_summary.py $(ls $O/pmc_f_$m/*.db | head -1) $(ls $O/pmc_w_$m/*.db | head -1) 4 $O/${m}_pmc.json $m > $O/${m}_pmc.md
done
C="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT"
hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak 2>/dev/null
timeout 300 rocprofv3 --kernel-trace --pmc $C -d $O/sq_cal -o r -- /tmp/mfma_peak > $O/sq_cal.log 2>&1
for m in infer train; do
  timeout 400 rocprofv3 --kernel-trace --pmc $C -d $O/sq_$m -o r -- python bench.py --mode $m --steps 1 --warmup 0 --no-cpu-baseline > $O/sq_$m.log 2>&1
  python tools/pmc_sq_summary.py $(ls $O/sq_$m/*.db | head -1) $(ls $O/sq_cal/*.db | head -1) $O/${m}_sq_pmc.json > $O/${m}_sq_pmc.md
done
unset VR_NO_SIDE_STREAM VR_NO_SPLIT_BATCH
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
if [ -z "$SKIP_TESTS" ]; then VR_MFMA_MODE=0 timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_mfma_mode0.log 2>&1; echo "pytest (VR_MFMA_MODE=0) rc=$?"; grep -n "passed\|failed" $O/pytest_mfma_mode0.log | tail -2; fi
find $O -name "*.db" -delete; rm -rf $O/kt_* $O/ktc_infer $O/ktc_train $O/pmc_f_* $O/pmc_w_* $O/sq_cal $O/sq_infer $O/sq_train 2>/dev/null
head -8 $O/train_kernel_trace.md; tail -1 $O/train_kernel_trace.md; head -3 $O/train_pmc.md; head -3 $O/infer_pmc.md
cat $O/infer_timeline_concurrent.txt | head -8; cat $O/train_timeline_concurrent.txt | head -8
grep "all kernels" $O/*_sq_pmc.md
