#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for d in 0 1 2 8 16 24 32 3; do echo -n "VR_WW_DBG=$d  "; VR_WW_DBG=$d timeout 300 python bench.py --mode train --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], j['ms_per_step'], j['roofline']['kernel_ms_per_step'])"; done
