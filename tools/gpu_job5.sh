#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/j5; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -30 $O/pytest.log
