#!/bin/bash
# round 6, GPU call 18: why are the decoder weight gradients (dec1/dec2/dec3: 103-120 TF) slower per chunk than the encoder ones (167-183 TF)?  slab-count sweep, per-layer times
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call18; rm -rf $O; mkdir -p $O
for pt in 512 768 1024 2048; do
VR_WW_PTARGET=$pt VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 3 --warmup 2 --no-cpu-baseline > $O/bench_$pt.json 2> $O/dump_$pt.txt
python - $O/dump_$pt.txt $pt $O/bench_$pt.json <<'PY'
import re,sys,json
rows=[ln for ln in open(sys.argv[1]) if '[vr-prof]' in ln and 'wgrad_wino' in ln]
k=len(rows)//3
sel={}
for ln in rows[-k:]:
    m=re.match(r'\[vr-prof\] (.*?) +([\d.]+) us +([\d.]+) GFLOP', ln)
    body=m.group(1); tag=body[58:].strip()
    for key in ('stg3_full_band_net.dec1','stg3_full_band_net.dec2','stg3_full_band_net.dec3','stg3_full_band_net.dec4','stg3_full_band_net.enc2.conv2','stg3_full_band_net.enc3.conv2','stg2_low_band_net.0.dec2','stg3_full_band_net.enc1'):
        if key in tag: sel[key]=(float(m.group(2)), float(m.group(3)))
tot=sum(float(re.match(r'\[vr-prof\] (.*?) +([\d.]+) us', ln).group(2)) for ln in rows[-k:])
j=json.loads(open(sys.argv[3]).read().splitlines()[-1])
print('PTARGET', sys.argv[2], 'step %.2f ms' % j['ms_per_step'], 'wgrad_wino total %.2f ms |' % (tot/1e3), ' '.join('%s %.0fus %.0fTF' % (k.split('.')[-2 if 'conv2' in k else -1] if False else k.replace('stg3_full_band_net.','s3.').replace('stg2_low_band_net.0.','s2l.'), v[0], v[1]/v[0]*1e3) for k,v in sel.items()))
PY
done
