#!/bin/bash
# round 5, GPU session M (library 42970108...): both columns of the mutant table once more on the final source (lib/mut_*.so,
# lib/mut_*_audit.so built from it); knob sweep at config 5's shape (tile shapes, group sizes, the LDS-staged variant), interleaved
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05m; mkdir -p $O gpurun_out/mutants; rm -f gpurun_out/mutants/results.json gpurun_out/mutants/audit.json
AB_EXTRA="--no-worst-case --no-parity-check --no-train-leg --size 512 --lights 18 --samples 320 --faces 1" AB_STEPS=300 timeout 900 tools/ab.sh default tile_w=8 tile_w=32 group=2 lds_stage=1 > $O/config5_knob_sweep.txt 2>&1; cat $O/config5_knob_sweep.txt
timeout 1500 python tools/mutants.py run > $O/mutants_run.log 2>&1; grep -c killed $O/mutants_run.log; grep SURVIVED $O/mutants_run.log | cut -c1-80
timeout 900 python tools/mutants.py run-audit > $O/mutants_audit.log 2>&1; grep -c '"violations": 0,' $O/mutants_audit.log
