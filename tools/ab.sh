#!/bin/bash
# A/B sweep of gcfr_tune settings on the bench workload (run on the GPU box).  Interleaved, 3 rounds.
# usage: tools/ab.sh "cfg1" "cfg2" ...   (cfg = comma list key=value, "direct" = no-workspace kernel,
#        "lib:<file>" = an alternative build under geomconsistentfr_amd/lib/; AB_EXTRA = extra bench flags)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
CFGS=("$@")
[ ${#CFGS[@]} -eq 0 ] && CFGS=("0=32" "0=16" "0=8" "0=64" "0=32,1=1" "direct")
for round in 1 2 3; do
  for cfg in "${CFGS[@]}"; do
    unset GCFR_HIP_LIB
    if [ "$cfg" = "direct" ]; then extra="--direct";
    elif [[ "$cfg" == lib:* ]]; then export GCFR_HIP_LIB="$REPO/geomconsistentfr_amd/lib/${cfg#lib:}"; extra="";
    else extra="--tune $cfg"; fi
    python bench.py $AB_EXTRA --no-cpu-baseline --steps ${AB_STEPS:-100} $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$cfg'.ljust(12), 'step %.1f G/s' % (d['value']/1e9), 'kernel+prepass %.4f ms  %.1f G/s' % (d['roofline']['avg_launch_ms'], d['roofline']['kernel_ray_steps_per_sec']/1e9))"
  done
done
