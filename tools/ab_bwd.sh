#!/bin/bash
# A/B of library builds on the fused backward kernel: tools/ab_bwd.sh <lib.so | default> ...
# per build: kernel-trace average of render_bwd_single_light_kernel on tools/bwd_bench.py (dense upstream gradient, B = 32),
# its HBM traffic (FETCH_SIZE / WRITE_SIZE, one PMC pass each; KiB, read side x2 on gfx950), and the in-step duration
# bench.py --workload train reports (masked upstream gradient).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  unset GCFR_HIP_LIB
  [ "$t" != "default" ] && export GCFR_HIP_LIB=$REPO/geomconsistentfr_amd/lib/$t
  rm -rf /tmp/abb
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abb/ks -o t -- python $REPO/tools/bwd_bench.py --iters 20 > /dev/null 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/abb/$c -o p -- python $REPO/tools/bwd_bench.py --iters 10 > /dev/null 2>&1
  done
  python - "$t" <<'PY'
import csv, glob, sys, collections
avg = {}
for f in glob.glob('/tmp/abb/ks/**/*kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'render_bwd' in r['Name'] or 'shadow_fwd_quad' in r['Name']:
            avg[r['Name'].split('(')[0][-48:]] = (float(r['AverageNs']) / 1e3, int(r['Calls']))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/abb/*SIZE/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'render_bwd' in row['Kernel_Name']:
            acc[row['Kernel_Name'].split('(')[0][-48:]][row['Counter_Name']].append(float(row['Counter_Value']))
print('==', sys.argv[1])
for k, (us, n) in avg.items():
    c = acc.get(k, {})
    fe = sum(c['FETCH_SIZE']) / len(c['FETCH_SIZE']) if c.get('FETCH_SIZE') else 0.0
    wr = sum(c['WRITE_SIZE']) / len(c['WRITE_SIZE']) if c.get('WRITE_SIZE') else 0.0
    print('   %-50s avg %.1f us (%d calls)   HBM %.1f MB / launch (fetch %.0f KiB x2, write %.0f KiB)' % (k, us, n, (2 * fe + wr) * 1024 / 1e6, fe, wr))
PY
  python $REPO/bench.py --workload train --steps 10 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   train step %.2f ms;' % d['ms_per_step'], 'in-step render block:', d['render_block_ms']['forward_march_kernel'], d['render_block_ms']['fused_backward_kernel'])"
done
