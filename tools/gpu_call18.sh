#!/bin/bash
# round 5, GPU call 18: the whole GPU suite + smoke on the final tree
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/call18; rm -rf $O; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -n "passed\|failed" $O/pytest.log | tail -2
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
