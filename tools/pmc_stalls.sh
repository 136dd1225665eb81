REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pq
i=0
for pass in "TA_TA_BUSY_sum TA_BUSY_avr TA_BUSY_max GRBM_GUI_ACTIVE" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TA_BUFFER_WAVEFRONTS_sum TA_BUFFER_READ_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SMEM SQ_INST_LEVEL_SMEM" "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAIT_IFETCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $pass --output-format csv -d /tmp/pq/p$i -o pq -- python $REPO/bench.py --no-cpu-baseline --steps 10 --warmup 2 --streams 1 --no-graph --regions 1 --no-worst-case --faces 128 > /tmp/pq_$i.log 2>&1 || tail -2 /tmp/pq_$i.log
done
python - <<'PY'
import csv, glob, collections
acc=collections.defaultdict(list)
for f in glob.glob('/tmp/pq/**/*counter_collection.csv', recursive=True):
    for row in csv.DictReader(open(f)):
        if 'shadow_fwd_quad' in row['Kernel_Name']:
            acc[row['Counter_Name']].append(float(row['Counter_Value']))
for k, v in sorted(acc.items()):
    print('%-40s %.4g' % (k, sum(v)/len(v)))
PY
