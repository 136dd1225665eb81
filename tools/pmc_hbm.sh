#!/bin/bash
# HBM traffic of the march kernel per launch (FETCH_SIZE / WRITE_SIZE, own PMC pass, no tracing): tools/pmc_hbm.sh [lib.so] [bench flags...]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
if [ -n "$1" ] && [[ "$1" == *.so ]]; then export GCFR_HIP_LIB=$REPO/geomconsistentfr_amd/lib/$1; shift; fi
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ph
for c in FETCH_SIZE WRITE_SIZE; do   # one counter per pass, un-overlapped launches (as tools/prof.sh)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d /tmp/ph/$c -o ph -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 --streams 1 --no-graph --regions 1 --no-worst-case "$@" > /dev/null 2>&1
done
python - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/ph/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        acc[row['Kernel_Name'].split('(')[0][-70:]][row['Counter_Name']].append(float(row['Counter_Value']))
for k, c in acc.items():
    fe, wr = (sum(c[n]) / len(c[n]) if c[n] else 0.0 for n in ('FETCH_SIZE', 'WRITE_SIZE'))
    # KiB; gfx950 FETCH_SIZE tallies 128-B requests at 64 B: read side doubled (MI355X_MICROARCH.md, HBM)
    print('%-72s fetch %8.1f KiB  write %8.1f KiB  -> HBM %.1f MB / launch' % (k, fe, wr, (2 * fe + wr) * 1024 / 1e6))
PY
