#!/bin/bash
# Side measurements next to the headline bench (run on the GPU box): batch scaling, worst-case mask,
# config 5 shape.  Prints one compact line per run.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
run() {
  python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); c=d['config']
print('B=%-3d %dx%d L=%-2d N=%-3d %-55s step %.3f ms  %.1f G ray-steps/s  %.0f faces/s  | march %.3f ms %.1f G/s' % (c['faces_per_gpu'], c['H'], c['W'], c['lights_per_face'], c['n_samples'], ' '.join(sys.argv[1:]), d['ms_per_step'], d['value']/1e9, d['faces_per_sec'], d['roofline']['avg_launch_ms'], d['roofline']['kernel_ray_steps_per_sec']/1e9))" "$@"
}
run --steps 1000
run --steps 1000 --streams 1
run --steps 1000 --streams 2
run --steps 1000 --no-graph
run --steps 300 --from-depth
run --steps 300 --mask ones
run --steps 300 --mask ones --streams 1
run --steps 100 --faces 32
run --steps 100 --faces 32 --streams 1
run --steps 60 --faces 128
run --steps 60 --faces 128 --streams 1
run --steps 60 --faces 128 --mask ones
run --steps 20 --faces 1 --size 512 --lights 18 --samples 320
run --steps 10 --faces 8 --size 512 --lights 18 --samples 320
run --steps 10 --faces 8 --size 512 --lights 18 --samples 320 --streams 1
run --steps 10 --faces 8 --size 512 --lights 18 --samples 320 --mask ones
run --steps 100 --faces 1
run --steps 100 --faces 1 --streams 1
