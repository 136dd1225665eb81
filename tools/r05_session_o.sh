#!/bin/bash
# round 5, GPU session O (library 42970108..., host change only): the normals stage as its own launch from 8 lights per face on
# (block.normals_stage_for).  Suite incl. its bit-identity test, the crossover (4 / 8 / 18 lights per 512 x 512 face, fused against own
# launch, interleaved), the driver's command, config 5 stand-alone
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05o; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -2
for round in 1 2; do for L in 4 8 18; do for ns in fused kernel; do
  python bench.py --no-cpu-baseline --no-worst-case --no-parity-check --no-train-leg --size 512 --lights $L --samples 320 --faces 1 --steps 300 --normals-stage $ns 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_stream') or {}; print('lights $L normals-stage $ns'.ljust(34), 'step %.1f G/s' % (d['value']/1e9), ' %.4f ms/step' % d['ms_per_step'], ' 1-stream %.1f G/s' % (s.get('ray_steps_per_sec',0)/1e9))"
done; done; done > $O/normals_stage_crossover.txt 2>&1; cat $O/normals_stage_crossover.txt
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.json; echo
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --size 512 --lights 18 --samples 320 --faces 1 --steps 300 > $O/bench_config5.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_config5.json').read().strip().splitlines()[-1]); print('bench_config5', d['value'], d['ms_per_step'])"
