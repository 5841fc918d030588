#!/bin/bash
# round 6, GPU call 19: slab count of the Winograd weight gradient from the dispatch geometry (new default) vs 512 workgroups (VR_WW_PTARGET=512)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call19; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_train.py tests/test_gpu_b16.py -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -2 $O/pytest.log
for pt in 512 0 512 0; do
VR_WW_PTARGET=$pt VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$pt.json 2> $O/dump_$pt.txt
python - $O/dump_$pt.txt $pt $O/bench_$pt.json <<'PY'
import re,sys,json
rows=[ln for ln in open(sys.argv[1]) if '[vr-prof]' in ln and ('wgrad_wino' in ln or 'wgrad_reduce' in ln)]
k=len(rows)//3
sel={}
tot=0.0; red=0.0
for ln in rows[-k:]:
    m=re.match(r'\[vr-prof\] (.*?) +([\d.]+) us +([\d.]+) GFLOP', ln)
    body=m.group(1); tag=body[58:].strip()
    if 'wgrad_reduce' in body: red+=float(m.group(2)); continue
    tot+=float(m.group(2))
    for key in ('stg3_full_band_net.dec1','stg3_full_band_net.dec2','stg3_full_band_net.dec3','stg3_full_band_net.dec4','stg3_full_band_net.enc2.conv2','stg3_full_band_net.enc3.conv2','stg2_low_band_net.0.dec2','stg2_low_band_net.0.dec3','stg3_full_band_net.enc1'):
        if key in tag: sel[key]=(float(m.group(2)), float(m.group(3)))
j=json.loads(open(sys.argv[3]).read().splitlines()[-1])
print('PTARGET', sys.argv[2], 'step %.2f ms' % j['ms_per_step'], 'wgrad_wino %.2f ms, slab sum %.2f |' % (tot/1e3, red/1e3), ' '.join('%s %.0fus' % (k.replace('stg3_full_band_net.','s3.').replace('stg2_low_band_net.0.','s2l.'), v[0]) for k,v in sel.items()))
PY
done
