#!/bin/bash
# round 6, GPU call 3: conv_x3pp (DMA staging) correctness on the small shapes + three layers; pp by dbg (32 = s_setprio over the multiply phase)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call3; rm -rf $O; mkdir -p $O
timeout 300 tools/_build/x3pp_proto 1 > $O/proto_correct.txt 2>&1; echo "correctness rc=$?"
python - <<'PY'
import re
for ln in open('gpurun_out/r6call3/proto_correct.txt'):
    items=re.findall(r'(pp\d|h<\d+,\d+>)\s+[\d.]+ us\s+\d+ TF ([\d.e+-]+|inf|nan)', ln)
    print(ln[:40], ' '.join('%s:%s'%i for i in items))
PY
timeout 300 tools/_build/x3pp_proto 2 > $O/proto_m2.txt 2>&1; cut -c1-400 $O/proto_m2.txt
timeout 300 tools/_build/x3pp_proto 2 32 > $O/proto_m2_prio.txt 2>&1; cut -c1-400 $O/proto_m2_prio.txt
