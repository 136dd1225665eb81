#!/bin/bash
# round 5, GPU session G (the final library once more, after the rough loop went to three samples in flight): the -m gpu suite, both
# columns of the mutant table, profiles (kernel trace + PMC passes), work counts, soaks, the audit (product + every march unit), the
# interleaved A/B against round 4's fixed-cost form, bench lines.  The driver's command runs afterwards (session H), when
# profiles/pmc_summary.json has been rebuilt from this session's counters.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05g; mkdir -p $O gpurun_out/mutants; [ -z "$SKIP_MUTANTS" ] && rm -f gpurun_out/mutants/results.json gpurun_out/mutants/audit.json
# (SKIP_MUTANTS=1: everything but the mutant table, which needs its 54 libraries built -- tools/mutants.py build, build-audit)
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -2
[ -z "$SKIP_MUTANTS" ] && timeout 1500 python tools/mutants.py run > $O/mutants_run.log 2>&1; grep -c killed $O/mutants_run.log; grep SURVIVED $O/mutants_run.log | cut -c1-80
[ -z "$SKIP_MUTANTS" ] && timeout 900 python tools/mutants.py run-audit > $O/mutants_audit.log 2>&1; grep -c '"violations": 0,' $O/mutants_audit.log
tools/prof.sh r05_fwd fwd > $O/prof_fwd.log 2>&1
tools/prof.sh r05_fwd128 fwd --faces 128 > $O/prof_fwd128.log 2>&1
tools/prof.sh r05_bwd bwd > $O/prof_bwd.log 2>&1
export GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/count.so
python tools/count_work.py --out $O/work_counts.json > /dev/null 2>&1
python tools/count_work.py --mask ones --out $O/work_counts_ones.json > /dev/null 2>&1
python tools/count_work.py --depth-noise 400 --out $O/work_counts_noise400.json > /dev/null 2>&1
python tools/count_work.py --faces 128 --out $O/work_counts_b128.json > /dev/null 2>&1
unset GCFR_HIP_LIB
timeout 600 python tools/soak_parity.py --cases 10000 --seed 81 > $O/soak_argmin.json 2> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 10000 --seed 82 --no-argmin > $O/soak_noargmin.json 2>> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 2400 --seed 83 --tune pixels=1 > $O/soak_pixels.json 2>> $O/soak.err
timeout 900 python tools/soak_parity.py --config5 64 --seed 84 --no-argmin > $O/soak_config5_noargmin.json 2>> $O/soak.err
timeout 900 python tools/soak_backward.py --cases 3000 --seed 86 > $O/soak_backward.json 2>> $O/soak.err
cat $O/soak_*.json | cut -c1-260
GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit.so python tools/audit.py --random 4000 --family-seeds 16 --more facets=120,pits2=48 --seed 3 --out $O/audit_product.json | cut -c1-420
bash tools/r05_audit_matrix.sh 2>&1 | tail -1
AB_EXTRA="--no-worst-case --no-parity-check --no-train-leg" AB_STEPS=3000 timeout 900 tools/ab.sh default lib:r04_fixed.so > $O/fixed_cost_ab.txt 2>&1; cat $O/fixed_cost_ab.txt
python bench.py --no-cpu-baseline --no-train-leg > $O/bench_default_3000.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --data ffhq > $O/bench_ffhq.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --size 512 --lights 18 --samples 320 --faces 1 --steps 300 > $O/bench_config5.json 2>/dev/null
python bench.py --workload train --steps 20 > $O/bench_train.json 2>/dev/null
for f in bench_default_3000 bench_ffhq bench_config5 bench_train; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
