#!/bin/bash
# quick PMC comparison of builds / knob settings on the march kernel: tools/pmc_quick.sh <faces> <cfg>...   cfg = "default" | "lib:<file>" | "<knobs>"
REPO=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  unset GCFR_HIP_LIB; extra=""
  if [[ "$t" == lib:* ]]; then export GCFR_HIP_LIB=$REPO/geomconsistentfr_amd/lib/${t#lib:}; elif [ "$t" != "default" ]; then extra="--tune $t"; fi
  rm -rf /tmp/pq
  for pass in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY" \
              "SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC GRBM_GUI_ACTIVE" \
              "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_IFETCH"; do
    name=$(echo $pass | cut -c1-20 | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pq/$name -o pq -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 --streams 1 --no-graph --regions 1 --no-worst-case --faces $F $extra > /dev/null 2>&1
  done
  python - "$t" <<'PY'
import csv, glob, sys, collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'shadow_fwd_quad' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
print(sys.argv[1].ljust(12), {k: '%.4g' % (sum(v)/len(v)) for k, v in sorted(acc.items())})
PY
done
