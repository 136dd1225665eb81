#!/bin/bash
# round 5, GPU session Q: HEAD as the round leaves it -- the whole -m gpu suite and the smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05q; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
