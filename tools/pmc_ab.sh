#!/bin/bash
# PMC comparison of knob settings / builds on the march kernel, stall side included:
#   tools/pmc_ab.sh <faces> <cfg>...      cfg = "default" | "lib:<file>[:knobs]" | "<knobs>"   (PMC_EXTRA = extra bench flags)
# One rocprofv3 --pmc pass per counter group (never combined with a trace domain), un-overlapped launches.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
F=$1; shift
cd /tmp && export TMPDIR=/tmp
for t in "$@"; do
  unset GCFR_HIP_LIB; extra=""
  if [[ "$t" == lib:* ]]; then rest="${t#lib:}"; export GCFR_HIP_LIB=$REPO/geomconsistentfr_amd/lib/${rest%%:*}; [[ "$rest" == *:* ]] && extra="--tune ${rest#*:}";
  elif [ "$t" != "default" ]; then extra="--tune $t"; fi
  rm -rf /tmp/pq
  i=0
  for pass in "SQ_INSTS_VALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD SQ_INSTS_LDS" \
              "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" \
              "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INSTS_SMEM SQ_BUSY_CYCLES" \
              "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pq/p$i -o pq -- python $REPO/bench.py --no-cpu-baseline --no-worst-case --steps 10 --warmup 2 --streams 1 --no-graph --regions 1 --no-worst-case --faces $F $PMC_EXTRA $extra > /tmp/pq_$i.log 2>&1 || tail -2 /tmp/pq_$i.log
  done
  python - "$t" <<'PY'
import csv, glob, sys, collections
acc = collections.defaultdict(list)
name = None
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'shadow_fwd_quad' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
            name = row['Kernel_Name'].split('(')[0]
m = {k: sum(v) / len(v) for k, v in acc.items()}
print('== %s   (%s)' % (sys.argv[1], name))
wc = m.get('SQ_WAVE_CYCLES', 0) or 1
gui = m.get('GRBM_GUI_ACTIVE', 0) / 8.0 or 1
print('   VALU insts %.3fM  VMEM_RD %.3fM  LDS insts %.3fM  waves %d' % (m.get('SQ_INSTS_VALU', 0) / 1e6, m.get('SQ_INSTS_VMEM_RD', 0) / 1e6, m.get('SQ_INSTS_LDS', 0) / 1e6, m.get('SQ_WAVES', 0)))
print('   wave time: s_waitcnt %.1f %%  issue-stall %.1f %%  issuing %.1f %%   (SQ_WAVE_CYCLES %.4g quad-cycles)' % (100 * m.get('SQ_WAIT_ANY', 0) / wc, 100 * m.get('SQ_WAIT_INST_ANY', 0) / wc, 100 * m.get('SQ_ACTIVE_INST_ANY', 0) / wc, wc))
print('   TA busy avg %.1f %%  max %.1f %%  (kernel %.4g cycles per XCD)   L1 accesses %.3fM  L2 reads %.3fM' % (100 * m.get('TA_BUSY_avr', 0) / gui, 100 * m.get('TA_BUSY_max', 0) / gui, gui, m.get('TCP_TOTAL_CACHE_ACCESSES_sum', 0) / 1e6, m.get('TCP_TCC_READ_REQ_sum', 0) / 1e6))
print('   LDS: idx-active %.4g  bank-conflict %.4g  wait-inst-lds %.4g quad-cycles' % (m.get('SQ_LDS_IDX_ACTIVE', 0), m.get('SQ_LDS_BANK_CONFLICT', 0), m.get('SQ_WAIT_INST_LDS', 0)))
PY
done
