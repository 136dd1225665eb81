#!/bin/bash
# quick PMC comparison of gcfr_options knob settings (bench.py --tune): tools/pmc_quick.sh <faces> "<tune A>" "<tune B>" ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  rm -rf /tmp/pq; rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d /tmp/pq -o pq -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 --faces $F --tune "$t" > /dev/null 2>&1
  python - "$t" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'shadow_fwd_quad' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
print(sys.argv[1].ljust(8), {k: '%.3g' % (sum(v)/len(v)) for k, v in sorted(acc.items())})
PY
done
