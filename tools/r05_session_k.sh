#!/bin/bash
# round 5, GPU session K: statistics chunks of 32,768 pixels for images of 9 ... 16 plain chunks (512 x 512: eight records instead of
# sixteen, folded on the scalar unit by every march wave instead of seven DPP reductions).  lib/prev.so = the library before the change,
# lib/new.so = with it (both -DGCFR_FAST_BUILD); the product library is the new source.  Interleaved A/B at config 5's shape and on the
# bench faces, then the whole evidence set on the new product: suite, soaks, audits, kernel trace + PMC passes, bench lines.
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05k; mkdir -p $O
AB_EXTRA="--no-worst-case --no-parity-check --no-train-leg --size 512 --lights 18 --samples 320 --faces 1" AB_STEPS=300 timeout 900 tools/ab.sh lib:prev.so lib:new.so > $O/stats_chunk_ab_config5.txt 2>&1; cat $O/stats_chunk_ab_config5.txt
AB_EXTRA="--no-worst-case --no-parity-check --no-train-leg" AB_STEPS=3000 timeout 900 tools/ab.sh lib:prev.so lib:new.so > $O/stats_chunk_ab_bench.txt 2>&1; cat $O/stats_chunk_ab_bench.txt
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 600 python tools/soak_parity.py --cases 10000 --seed 101 > $O/soak_argmin.json 2> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 10000 --seed 102 --no-argmin > $O/soak_noargmin.json 2>> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 2400 --seed 103 --tune pixels=1 > $O/soak_pixels.json 2>> $O/soak.err
timeout 900 python tools/soak_parity.py --config5 64 --seed 104 --no-argmin > $O/soak_config5_noargmin.json 2>> $O/soak.err
timeout 900 python tools/soak_parity.py --config5 64 --seed 105 > $O/soak_config5_argmin.json 2>> $O/soak.err
timeout 900 python tools/soak_backward.py --cases 3000 --seed 106 > $O/soak_backward.json 2>> $O/soak.err
cat $O/soak_*.json | cut -c1-260
timeout 300 python tools/big_size_check.py > $O/big_size_check.log 2>&1; tail -3 $O/big_size_check.log
GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit.so timeout 900 python tools/audit.py --random 4000 --family-seeds 16 --more facets=120,pits2=48 --seed 5 --out $O/audit_product.json | cut -c1-420
GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit.so timeout 900 python tools/audit.py --random 0 --families none --config5 16 --seed 12 --out $O/audit_config5.json | cut -c1-420
tools/prof.sh r05_fwd fwd > $O/prof_fwd.log 2>&1
tools/prof.sh r05_fwd128 fwd --faces 128 > $O/prof_fwd128.log 2>&1
tools/prof.sh r05_bwd bwd > $O/prof_bwd.log 2>&1
cd $GRAFT_REPO_ROOT
python bench.py --no-cpu-baseline --no-train-leg > $O/bench_default_3000.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --data ffhq > $O/bench_ffhq.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --size 512 --lights 18 --samples 320 --faces 1 --steps 300 > $O/bench_config5.json 2>/dev/null
for f in bench_default_3000 bench_ffhq bench_config5; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
# keep the merged output small: the PMC passes' raw csv files are what summarize_profile.py reads
du -sh gpurun_out | tail -1
