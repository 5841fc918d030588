#!/bin/bash
# round 6, GPU call 27: the row-batched guarded epilogue (shipped) against the per-element one (tools/_build/libvr_old_epilogue.so, built from 0cc202b~1's conv_epilogue.h) -- per-launch times of the data gradients that run through it
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call27; rm -rf $O; mkdir -p $O
for v in new old new old; do
  if [ $v = old ]; then export VR_LIB_PATH=$PWD/tools/_build/libvr_old_epilogue.so; else unset VR_LIB_PATH; fi
  VR_PROFILE_DUMP=1 timeout 300 python bench.py --mode train --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_$v.json 2> $O/dump_$v.txt
  python - $O/dump_$v.txt $v $O/bench_$v.json <<'PY'
import re,sys,json
rows=[ln for ln in open(sys.argv[1]) if '[vr-prof]' in ln]
k=len(rows)//3
sel={}
tot=0
for ln in rows[-k:]:
    m=re.match(r'\[vr-prof\] (.*?) +([\d.]+) us', ln)
    body=m.group(1); tag=body[58:].strip(); tot+=float(m.group(2))
    for key in ('dgrad stg3_full_band_net.dec1.conv1','dgrad stg3_full_band_net.enc1','dgrad stg2_low_band_net.0.dec1','dgrad stg3_full_band_net.dec2','dgrad stg2_low_band_net.0.enc1','dgrad stg3_full_band_net.dec4'):
        if tag.startswith(key): sel[key.replace('dgrad ','').replace('_full_band_net','').replace('_low_band_net.0','l')]=float(m.group(2))
j=json.loads(open(sys.argv[3]).read().splitlines()[-1])
print(sys.argv[2], 'step %.2f ms serial %.2f |' % (j['ms_per_step'], tot/1e3), ' '.join('%s %.0f' % kv for kv in sel.items()))
PY
done
