#!/bin/bash
# kernel-trace stats of one command, condensed: tools/kstats.sh <command...>   (default: bench.py --steps 50)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
CMD="$*"
[ -z "$CMD" ] && CMD="python $REPO/bench.py --steps 50 --warmup 3 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o trace -- $CMD > /dev/null 2>&1
python - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ks/**/trace_kernel_stats.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'gcfr' in r['Name']:
            print(r['Name'].split('(')[0][-60:].ljust(60), r['Calls'].rjust(5), 'avg %.1f us' % (float(r['AverageNs'])/1e3), 'min %.1f' % (float(r['MinNs'])/1e3), 'max %.1f' % (float(r['MaxNs'])/1e3))
PY
