#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + separate PMC passes (never combined with a
# trace domain).  usage: tools/prof.sh <tag> [fwd|bwd] [extra bench flags...]  -> writes gpurun_out/prof_<tag>/...
set -u
TAG=${1:-r02}
WHAT=${2:-fwd}
shift; shift
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$WHAT" = "bwd" ]; then
  CMD="python $REPO/tools/bwd_bench.py --iters 10 $*"
else
  CMD="python $REPO/bench.py --steps 20 --warmup 3 --no-cpu-baseline --streams 1 --no-graph --regions 1 --no-worst-case $*"   # un-overlapped launches
fi
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o trace -- $CMD > $OUT/trace.log 2>&1
for pass in "FETCH_SIZE" "WRITE_SIZE" \
            "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY" \
            "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
            "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" \
            "SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE" \
            "SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32" \
            "SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_SALU"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass --output-format csv -d $OUT/pmc_$name -o pmc -- $CMD > $OUT/pmc_$name.log 2>&1
done
ls -R $OUT | head -60
