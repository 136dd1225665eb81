#!/bin/bash
# Is the four-in-flight rate bimodal?  (round 5: variants of the march at lower occupancy showed 1.24 T or 2.18 T from run to run.)
# usage (GPU box): tools/exp_modes.sh <runs> "<lib or default>[:ENV=VALUE...]" ...   -> one line per run
REPO=${GRAFT_REPO_ROOT:-/root/repo}; cd $REPO
RUNS=$1; shift
for cfg in "$@"; do
  lib="${cfg%%:*}"; envs=""; [[ "$cfg" == *:* ]] && envs="${cfg#*:}"
  for i in $(seq 1 $RUNS); do
    ( unset GCFR_HIP_LIB; [ "$lib" != "default" ] && export GCFR_HIP_LIB="$REPO/geomconsistentfr_amd/lib/$lib"
      for e in ${envs//:/ }; do export "$e"; done
      python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --no-parity-check --steps ${MODE_STEPS:-1000} $MODE_EXTRA 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_stream') or {}; print('$cfg'.ljust(40), 'step %.1f G/s' % (d['value']/1e9), ' 1-stream %.1f G/s' % (s.get('ray_steps_per_sec',0)/1e9), ' ratio %.2f' % (d['value']/max(1.0,s.get('ray_steps_per_sec',1))))" )
  done
done
