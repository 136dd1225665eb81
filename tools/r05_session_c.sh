#!/bin/bash
# round 5, GPU session C: the fixed-cost trims after the report (stencil loads, branch-free slabs): tests, short soak, interleaved A/B
# against the library of commit 4b09bed (lib/r05_head.so), and which job of the prepass sets its duration (tools/prepass_jobs.py)
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05c; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python tools/soak_parity.py --cases 3000 --seed 61 > $O/soak_argmin.json 2> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 3000 --seed 62 --no-argmin > $O/soak_noargmin.json 2>> $O/soak.err
cat $O/soak_*.json | cut -c1-300
AB_STEPS=3000 AB_EXTRA="--no-worst-case --no-train-leg" timeout 1200 tools/ab.sh "lib:r05_head.so" "default" > $O/ab.txt 2>&1; cat $O/ab.txt
timeout 900 python tools/prepass_jobs.py run > $O/prepass_jobs.log 2>&1; cat $O/prepass_jobs.log | cut -c1-200
cp gpurun_out/prepass_jobs.json $O/ 2>/dev/null
