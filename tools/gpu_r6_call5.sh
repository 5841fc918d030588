#!/bin/bash
# round 6, GPU call 5: kernel-coverage test, the new edge-read test, quick parity subset after the conv_x3h neighbour clamp
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call5; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_kernel_coverage.py -m gpu -q -p no:cacheprovider -x -s > $O/coverage.log 2>&1; echo "coverage rc=$?"; tail -95 $O/coverage.log | cut -c1-200
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "upsample or forward_taps or predict_mask_full" > $O/parity.log 2>&1; echo "parity rc=$?"; tail -5 $O/parity.log
