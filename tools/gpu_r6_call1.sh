#!/bin/bash
# round 6, GPU call 1: conv_x3pp (ping-pong) against conv_x3h in the stand-alone harness -- correctness shapes, then the S30 inference layers
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call1; rm -rf $O; mkdir -p $O
timeout 300 tools/_build/x3pp_proto 1 > $O/proto_correct.txt 2>&1; echo "correctness rc=$?"
cat $O/proto_correct.txt | cut -c1-400
timeout 600 tools/_build/x3pp_proto 0 > $O/proto_layers.txt 2>&1; echo "layers rc=$?"
grep -v "^small\|^one\|^two\|^3 \|^64ch\|^128 c\|^192 c\|^upsam\|^scale\|^subn\|^per-" $O/proto_layers.txt | cut -c1-420
