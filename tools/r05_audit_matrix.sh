#!/bin/bash
# round 5: the audit over every tile shape and group size, the LDS variant, the k-split and pixels = mask (lib/audit_full.so: an audit
# build of every march unit, tools/build_variant.sh audit_full -DGCFR_COUNTERS -DGCFR_AUDIT) -> gpurun_out/r05_audit/*.json + a total
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05_audit; mkdir -p $O; s=200
export GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit_full.so
for cfg in "tile_w=8,group=4" "tile_w=8,group=2" "tile_w=8,group=1" "tile_w=16,group=4" "tile_w=16,group=2" "tile_w=16,group=1" "tile_w=32,group=4" "tile_w=32,group=2" \
           "tile_w=64,group=4" "tile_w=64,group=1" "lds_stage=1" "ksplit=1"; do
  s=$((s+1)); f=$(echo $cfg | tr ',=' '__')
  timeout 600 python tools/audit.py --random 400 --family-seeds 4 --seed $s --tune $cfg --out $O/$f.json > /dev/null 2> $O/$f.err
done
python - <<'PY'
import glob, json
tot = {"depth_bound": 0, "termination": 0, "masked": 0}; viol = 0; use = 0.0
for f in sorted(glob.glob('gpurun_out/r05_audit/*.json')):
    d = json.load(open(f))
    for k in tot: tot[k] += d["claims_checked_lane_samples"][k]
    viol += d["violations"]; use = max(use, d["max_share_of_Kerr_used_by_a_bound_evaluation"])
    print(f.split('/')[-1], d["claims_checked_lane_samples"], d["violations"], d["max_share_of_Kerr_used_by_a_bound_evaluation"])
print('TOTAL', tot, 'violations', viol, 'max share of Kerr', use)
json.dump({"claims_checked_lane_samples": tot, "violations": viol, "max_share_of_Kerr_used_by_a_bound_evaluation": use,
           "configurations": sorted(f.split('/')[-1][:-5] for f in glob.glob('gpurun_out/r05_audit/*.json'))}, open('gpurun_out/r05_audit/TOTAL.json', 'w'), indent=1)
PY
