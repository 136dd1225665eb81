#!/bin/bash
# ONE parametrised GPU session runner (round 6: replaces the eighteen one-shot tools/r05_session_*.sh and the r05_*_matrix.sh /
# r05_ab_rough.sh scripts -- their outputs stay under profiles/, the recipes live here as named steps).
#
#   gpurun --timeout 1800 -- 'bash tools/session.sh <tag> <step> [<step> ...]'          outputs: gpurun_out/<tag>/
#
# steps (each bounded by its own `timeout`; a failing step is reported and the next one still runs):
#   suite            pytest -m gpu over tests/ (the whole GPU suite)              -> pytest_full.log
#   suite:<expr>     ... restricted with -k "<expr>" (use _ for spaces: suite:parity_or_configs)
#   smoke            __graft_entry__.smoke()                                      -> smoke.log
#   bench            the driver's command: bench.py --gpus 1 --steps 20 --warmup 5  -> bench_driver_cmd.json
#   bench_train      bench.py --workload train --steps 20                         -> bench_train.json
#   bench:<flags>    bench.py with the given flags (use _ for spaces)             -> bench_<flags>.json
#   gpus2            the N = 2 launch rehearsed on this box (--oversubscribe over gloo when it has one GPU), both workloads
#   profile          tools/prof.sh <tag> fwd + bwd + fwd128 (kernel trace + the PMC passes -> gpurun_out/prof_<tag>_{fwd,bwd,fwd128}, merged back
#                    by gpurun) and a first tools/summarize_profile.py <tag> for the log; profiles/ on the box does not travel back: run
#                    `python tools/summarize_profile.py <tag>` again at home on the merged directories, then tools/design_table.py --write
#   train_breakdown  rocprofv3 --kernel-trace of tools/train_breakdown.py run     -> train_trace/, train_phases.json, train_step_breakdown.md
#   audit            tools/audit.py on lib/audit.so (random 800 + every directed family)  -> audit_product.json
#   audit_matrix     the audit over every tile shape / group size / LDS / k-split (lib/audit_full.so: tools/build_variant.sh audit_full
#                    -DGCFR_COUNTERS -DGCFR_AUDIT)                                  -> audit_matrix/*.json, audit_matrix/TOTAL.json
#   soak             tools/soak_parity.py 10000 cases with and without argmin; tools/soak_backward.py 4000  -> soak_*.json
#   soak_matrix      the parity soak over every tile shape / group / LDS / k-split / bounds off / pixels = mask  -> soak_matrix/*.json
#   mutants[:n,n..]  tools/mutants.py run + run-audit (build / build-audit on the CPU first) for all or the named mutants
#   ab:<cfg>|<cfg>   tools/ab.sh interleaved A/B of knob settings / builds ("lib:x.so" entries), AB_EXTRA / AB_STEPS from the environment
cd ${GRAFT_REPO_ROOT:-/root/repo}
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
say() { echo "[session $TAG] $*"; }
for step in "$@"; do
  arg="${step#*:}"; [ "$arg" = "$step" ] && arg=""; arg="${arg//_/ }"; name="${step%%:*}"
  t0=$(date +%s)
  case $name in
    suite)
      if [ -n "$arg" ]; then timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider -k "$arg" > $O/pytest_k.log 2>&1; grep -n "passed\|failed" $O/pytest_k.log | tail -3
      else timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed\|^FAILED\|^ERROR" $O/pytest_full.log | tail -8; fi ;;
    smoke) timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log ;;
    bench)
      if [ -n "$arg" ]; then f=$(echo "$arg" | tr -c 'a-zA-Z0-9' '_' | cut -c1-60); timeout 900 python bench.py $arg > $O/bench_$f.json 2> $O/bench_$f.err; tail -c 400 $O/bench_$f.json; echo
      else timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.json; echo; fi ;;
    bench_train) timeout 900 python bench.py --workload train --steps 20 > $O/bench_train.json 2> $O/bench_train.err; tail -c 300 $O/bench_train.json; echo ;;
    gpus2)
      extra=""; [ "$(python -c 'import torch; print(torch.cuda.device_count())')" -lt 2 ] && extra="--oversubscribe"
      timeout 600 python bench.py --gpus 2 $extra --steps 200 --warmup 50 --no-cpu-baseline --no-worst-case --no-train-leg > $O/bench_gpus2.json 2> $O/bench_gpus2.err; tail -c 200 $O/bench_gpus2.json; echo
      timeout 600 python bench.py --gpus 2 $extra --workload train --steps 8 --warmup 4 > $O/bench_train_gpus2.json 2> $O/bench_train_gpus2.err; tail -c 200 $O/bench_train_gpus2.json; echo ;;
    profile)
      bash tools/prof.sh ${TAG}_fwd fwd > $O/prof_fwd.log 2>&1; bash tools/prof.sh ${TAG}_bwd bwd > $O/prof_bwd.log 2>&1
      bash tools/prof.sh ${TAG}_fwd128 fwd --faces 128 > $O/prof_fwd128.log 2>&1
      python tools/summarize_profile.py $TAG > $O/summarize.log 2>&1; tail -3 $O/summarize.log ;;
    train_breakdown)
      # (1) the phases without a profiler attached (events only): the step as the bench times it;  (2) the kernel trace of the same
      # run with a sentinel launch at every phase boundary -> by class, by phase, every kernel name.  Both for the step as it is
      # (SSIM blurs through ATen's depthwise kernels) and as rounds 2-5 ran it (--ssim-blur miopen): the "before" table.
      for kb in aten miopen; do
        timeout 600 python tools/train_breakdown.py run --steps 10 --ssim-blur $kb > $O/train_phases_noprof_$kb.log 2>&1; grep '^{' $O/train_phases_noprof_$kb.log | tail -1 > $O/train_phases_noprof_$kb.json
        ( cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tb && timeout 1200 rocprofv3 --kernel-trace --output-format csv -d /tmp/tb -o tb -- \
            python $OLDPWD/tools/train_breakdown.py run --steps 10 --phase-sentinels --ssim-blur $kb > /tmp/tb_run.log 2>&1 )
        grep '^{' /tmp/tb_run.log | tail -1 > $O/train_phases_$kb.json
        python tools/train_breakdown.py classify /tmp/tb --phases $O/train_phases_$kb.json --noprof $O/train_phases_noprof_$kb.json \
            --out $O/train_step_breakdown_$kb.md --json $O/train_step_kernels_$kb.json > /dev/null 2> $O/train_classify_$kb.err
        head -22 $O/train_step_breakdown_$kb.md; cat $O/train_phases_noprof_$kb.json; echo
      done ;;
    audit) GCFR_HIP_LIB=$PWD/geomconsistentfr_amd/lib/audit.so timeout 900 python tools/audit.py --random 800 --family-seeds 8 --more facets=60,pits2=24 --out $O/audit_product.json | tail -c 600; echo ;;
    audit_matrix)
      mkdir -p $O/audit_matrix; s=200; export GCFR_HIP_LIB=$PWD/geomconsistentfr_amd/lib/audit_full.so
      for cfg in "tile_w=8,group=4" "tile_w=8,group=2" "tile_w=8,group=1" "tile_w=16,group=4" "tile_w=16,group=2" "tile_w=16,group=1" "tile_w=32,group=4" \
                 "tile_w=32,group=2" "tile_w=64,group=4" "tile_w=64,group=1" "lds_stage=1" "ksplit=1"; do
        s=$((s+1)); f=$(echo $cfg | tr ',=' '__')
        timeout 600 python tools/audit.py --random 400 --family-seeds 4 --seed $s --tune $cfg --out $O/audit_matrix/$f.json > /dev/null 2> $O/audit_matrix/$f.err
      done; unset GCFR_HIP_LIB
      python - $O/audit_matrix <<'PY'
import glob, json, sys
d0 = sys.argv[1]; tot = {"depth_bound": 0, "termination": 0, "masked": 0}; viol = 0; use = 0.0
files = sorted(f for f in glob.glob(d0 + '/*.json') if not f.endswith('TOTAL.json'))
for f in files:
    d = json.load(open(f))
    for k in tot: tot[k] += d["claims_checked_lane_samples"][k]
    viol += d["violations"]; use = max(use, d["max_share_of_Kerr_used_by_a_bound_evaluation"])
print('TOTAL', tot, 'violations', viol, 'max share of Kerr', use)
json.dump({"claims_checked_lane_samples": tot, "violations": viol, "max_share_of_Kerr_used_by_a_bound_evaluation": use,
           "configurations": [f.split('/')[-1][:-5] for f in files]}, open(d0 + '/TOTAL.json', 'w'), indent=1)
PY
      ;;
    soak)
      timeout 900 python tools/soak_parity.py --cases 10000 --seed 606 > $O/soak_parity_argmin.json 2>/dev/null; tail -c 300 $O/soak_parity_argmin.json; echo
      timeout 900 python tools/soak_parity.py --cases 10000 --seed 607 --no-argmin > $O/soak_parity_noargmin.json 2>/dev/null; tail -c 300 $O/soak_parity_noargmin.json; echo
      timeout 900 python tools/soak_backward.py --cases 4000 --seed 608 > $O/soak_backward.json 2>/dev/null; tail -c 300 $O/soak_backward.json; echo ;;
    soak_matrix)
      mkdir -p $O/soak_matrix; s=100
      for cfg in "tile_w=8,group=4" "tile_w=8,group=2" "tile_w=8,group=1" "tile_w=16,group=4" "tile_w=16,group=2" "tile_w=16,group=1" "tile_w=32,group=4" "tile_w=64,group=4" \
                 "lds_stage=1,ksplit=0" "ksplit=1" "ksplit=0" "depth_bound_skip=0" "pixels=1,ksplit=0" "tile_w=32,group=2,depth_bound_skip=0"; do
        s=$((s+1)); f=$(echo $cfg | tr ',=' '__')
        python tools/soak_parity.py --cases 6400 --seed $s --tune $cfg > $O/soak_matrix/argmin_$f.json 2>/dev/null
        [[ "$cfg" == pixels* ]] || python tools/soak_parity.py --cases 3200 --seed $((s+50)) --tune $cfg --no-argmin > $O/soak_matrix/noargmin_$f.json 2>/dev/null
      done
      python - $O/soak_matrix <<'PY'
import glob, json, sys
tot = diff = 0
for f in sorted(glob.glob(sys.argv[1] + '/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    tot += d['pixels_compared']; diff += d['argmin_differences'] + d['lit_mask_mismatches'] + (1 if d['max_abs_err_min_dist'] else 0)
print('TOTAL pixels', tot, 'differences', diff)
PY
      ;;
    mutants)
      ns="${arg//,/ }"
      timeout 3000 python tools/mutants.py run $ns > $O/mutants_run.log 2>&1; tail -5 $O/mutants_run.log
      timeout 3000 python tools/mutants.py run-audit $ns > $O/mutants_audit.log 2>&1; tail -5 $O/mutants_audit.log
      cp gpurun_out/mutants/*.json $O/ 2>/dev/null ;;
    ab) IFS='|' read -ra CFGS <<< "${step#*:}"; timeout 1500 tools/ab.sh "${CFGS[@]}" > $O/ab.txt 2>&1; cat $O/ab.txt ;;
    *) say "unknown step $step" ;;
  esac
  say "$step: $(( $(date +%s) - t0 )) s"
done
