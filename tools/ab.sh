#!/bin/bash
# A/B sweep of gcfr_options knob settings (bench.py --tune) on the bench workload (run on the GPU box).
# Interleaved, 3 rounds.  usage: tools/ab.sh "cfg1" "cfg2" ...   (cfg = comma list knob=value, "direct" = no-workspace
# kernel, "lib:<file>" = an alternative build under geomconsistentfr_amd/lib/, "default" = no knobs;
# AB_EXTRA = extra bench flags, AB_STEPS = steps per run)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd $REPO
CFGS=("$@")
[ ${#CFGS[@]} -eq 0 ] && CFGS=("default" "tile_w=16" "tile_w=8,group=2" "lds_stage=0")
for round in 1 2 3; do
  for cfg in "${CFGS[@]}"; do
    unset GCFR_HIP_LIB
    if [ "$cfg" = "direct" ]; then extra="--direct";
    elif [ "$cfg" = "default" ]; then extra="";
    elif [[ "$cfg" == lib:* ]]; then rest="${cfg#lib:}"; export GCFR_HIP_LIB="$REPO/geomconsistentfr_amd/lib/${rest%%:*}"; extra=""; [[ "$rest" == *:* ]] && extra="--tune ${rest#*:}";
    else extra="--tune $cfg"; fi
    python bench.py $AB_EXTRA --no-cpu-baseline --steps ${AB_STEPS:-300} $extra 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); s=d.get('single_stream') or {}; print('$cfg'.ljust(34), 'step %.1f G/s' % (d['value']/1e9), ' march %.4f ms' % (d['roofline']['avg_launch_ms']), ' 1-stream %.1f G/s' % (s.get('ray_steps_per_sec',0)/1e9))"
  done
done
