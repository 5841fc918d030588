#!/bin/bash
# wgrad_x3h_kernel phase ablation (VR_WXH_DBG bits: 1 no loads, 4 no split pass, 8 no tile maxima); results are wrong on purpose
O=gpurun_out/wxh; mkdir -p $O
for d in 0 1 4 8 5 13; do
  VR_WXH_DBG=$d timeout 90 python bench.py --mode train --steps 3 --warmup 1 --no-cpu-baseline > $O/t$d.json 2> $O/t$d.err
done
python - <<'PY'
import json
for d in (0,1,4,8,5,13):
    try:
        j=json.loads(open('gpurun_out/wxh/t%d.json'%d).read()); r=j['roofline']
        row=[k for k in r['kernels'] if 'wgrad_x3h' in k[0]][0]
        print('dbg %2d: wgrad_x3h %.2f ms over %d launches; step %.2f ms' % (d,row[2],row[1],j['ms_per_step']))
    except Exception as e: print('dbg',d,'ERR',e)
PY
