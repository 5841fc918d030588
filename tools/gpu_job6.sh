#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j6; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -12 $O/pytest.log
for v in 0 1; do echo -n "VR_NO_TILED_STFT=$v "; if [ $v = 1 ]; then export VR_NO_TILED_STFT=1; fi; timeout 300 python bench.py --mode infer --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; done
unset VR_NO_TILED_STFT
export VR_NO_SIDE_STREAM=1 VR_NO_SPLIT_BATCH=1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt_infer -o r -- python bench.py --mode infer --steps 3 --warmup 1 --no-cpu-baseline > $O/kt_infer.log 2>&1
python tools/rocpd_summary.py $(ls $O/kt_infer/*.db | head -1) $O/r02_infer_kernel_trace_single_stream.md > /dev/null
timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pmc_f -o r -- python bench.py --mode infer --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_f.log 2>&1
timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pmc_w -o r -- python bench.py --mode infer --steps 1 --warmup 0 --no-cpu-baseline > $O/pmc_w.log 2>&1
python tools/pmc_summary.py $(ls $O/pmc_f/*.db | head -1) $(ls $O/pmc_w/*.db | head -1) 2 $O/r02_infer_pmc.json infer > $O/r02_infer_pmc.md
find $O -name "*.db" -size +20M -delete
grep -E "stft|mag_pad|apply_mask|coef|materialize|all kernels" $O/r02_infer_kernel_trace_single_stream.md
grep -E "stft|mag_pad|conv family" $O/r02_infer_pmc.md
