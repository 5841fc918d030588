#!/bin/bash
# round 6, GPU call 10: conv_x3d tests after the fixes, the whole GPU suite, the space-to-depth probe
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r6call10; rm -rf $O; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_x3d.py -m gpu -q -p no:cacheprovider -s > $O/pytest_x3d.log 2>&1; echo "pytest x3d rc=$?"; grep -E "passed|failed|error" $O/pytest_x3d.log | tail -3; grep -E "^FAILED|^ERROR|^E  " $O/pytest_x3d.log | head -20
timeout 600 python tools/s2d_probe.py > $O/s2d_probe.txt 2> $O/s2d_probe.err; echo "probe rc=$?"; cat $O/s2d_probe.txt; tail -3 $O/s2d_probe.err
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.log 2>&1; echo "pytest all rc=$?"; tail -8 $O/pytest_all.log
