#!/bin/bash
# round 5, GPU session A: the full -m gpu suite, the mutant table, the interleaved A/B of the fixed-cost cuts, the driver's bench command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O; rm -f gpurun_out/mutants/results.json
python -m pytest tests -m gpu -q > $O/pytest_full.log 2>&1; tail -6 $O/pytest_full.log | cut -c1-250
python tools/mutants.py run > $O/mutants_run.log 2>&1; tail -30 $O/mutants_run.log | cut -c1-250
AB_EXTRA="--no-worst-case --no-parity-check" AB_STEPS=3000 tools/ab.sh default lib:r04_fixed.so > $O/fixed_cost_ab.txt 2>&1; cat $O/fixed_cost_ab.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 600 $O/bench_driver_cmd.json
