#!/bin/bash
# round 6, GPU call 4: phase trace of conv_x3pp (one workgroup, cycles per phase body and barrier wait)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call4; rm -rf $O; mkdir -p $O
timeout 120 tools/_build/x3pp_proto 5 0 enc3b > $O/trace_enc3b.txt 2>&1; cut -c1-260 $O/trace_enc3b.txt
timeout 120 tools/_build/x3pp_proto 5 0 dec1 > $O/trace_dec1.txt 2>&1; cut -c1-260 $O/trace_dec1.txt
