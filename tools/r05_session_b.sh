#!/bin/bash
# round 5, GPU session B: profiles (kernel trace + PMC passes), work counts, soaks, the other bench lines, backward request census
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05; mkdir -p $O
tools/prof.sh r05_fwd fwd > $O/prof_fwd.log 2>&1
tools/prof.sh r05_fwd128 fwd --faces 128 > $O/prof_fwd128.log 2>&1
tools/prof.sh r05_bwd bwd > $O/prof_bwd.log 2>&1
export GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/count.so
python tools/count_work.py --out gpurun_out/r05/work_counts.json > /dev/null 2>&1
python tools/count_work.py --mask ones --out gpurun_out/r05/work_counts_ones.json > /dev/null 2>&1
python tools/count_work.py --depth-noise 400 --out gpurun_out/r05/work_counts_noise400.json > /dev/null 2>&1
python tools/count_work.py --faces 128 --out gpurun_out/r05/work_counts_b128.json > /dev/null 2>&1
export GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/bwd_count.so
python tools/bwd_requests.py > $O/bwd_requests.log 2>&1; cp gpurun_out/r05_bwd_requests.json $O/ 2>/dev/null
unset GCFR_HIP_LIB
python tools/soak_parity.py --cases 10000 --seed 51 > $O/soak_argmin.json 2> $O/soak.err
python tools/soak_parity.py --cases 10000 --seed 52 --no-argmin > $O/soak_noargmin.json 2>> $O/soak.err
python tools/soak_parity.py --cases 2400 --seed 53 --tune pixels=1 > $O/soak_pixels.json 2>> $O/soak.err
python tools/soak_parity.py --config5 64 --seed 54 > $O/soak_config5.json 2>> $O/soak.err
python tools/soak_backward.py --cases 3000 --seed 55 > $O/soak_backward.json 2>> $O/soak.err
cat $O/soak_*.json | cut -c1-400
python bench.py --no-cpu-baseline --no-train-leg > $O/bench_default_3000.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --data ffhq > $O/bench_ffhq.json 2>/dev/null
python bench.py --no-cpu-baseline --no-worst-case --size 512 --lights 18 --samples 320 --faces 1 --steps 300 > $O/bench_config5.json 2>/dev/null
python bench.py --workload train --steps 20 > $O/bench_train.json 2>/dev/null
python bench.py --workload train --steps 20 --pixels mask > $O/bench_train_pixels_mask.json 2>/dev/null
for f in bench_default_3000 bench_ffhq bench_config5 bench_train bench_train_pixels_mask; do python -c "
import json,sys; d=json.loads(open('$O/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"; done
