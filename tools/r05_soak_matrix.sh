#!/bin/bash
# round 5: the soak over every tile shape, group size, the LDS variant, the k-split, bounds off and pixels = mask on the final library
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_soak; mkdir -p $O; s=100
for cfg in "tile_w=8,group=4" "tile_w=8,group=2" "tile_w=8,group=1" "tile_w=16,group=4" "tile_w=16,group=2" "tile_w=16,group=1" "tile_w=32,group=4" "tile_w=64,group=4" \
           "lds_stage=1,ksplit=0" "ksplit=1" "ksplit=0" "depth_bound_skip=0" "pixels=1,ksplit=0" "tile_w=32,group=2,depth_bound_skip=0"; do
  s=$((s+1)); f=$(echo $cfg | tr ',=' '__')
  python tools/soak_parity.py --cases 6400 --seed $s --tune $cfg > $O/argmin_$f.json 2>/dev/null
  [[ "$cfg" == pixels* ]] || python tools/soak_parity.py --cases 3200 --seed $((s+50)) --tune $cfg --no-argmin > $O/noargmin_$f.json 2>/dev/null
done
python - <<'PY'
import glob, json
tot = diff = 0
for f in sorted(glob.glob('gpurun_out/r05_soak/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    tot += d['pixels_compared']; diff += d['argmin_differences'] + d['lit_mask_mismatches'] + (1 if d['max_abs_err_min_dist'] else 0)
    print(f.split('/')[-1], d['pixels_compared'], d['argmin_differences'], d['lit_mask_mismatches'], d['max_abs_err_min_dist'])
print('TOTAL pixels', tot, 'differences', diff)
PY
