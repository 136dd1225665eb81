#!/bin/bash
# round 5, GPU session H: the driver's command on the final library with the final PMC summary; the audit at config 5's shape
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05h; mkdir -p $O
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.json
GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit.so timeout 1200 python tools/audit.py --random 0 --families none --config5 16 --seed 11 --out $O/audit_config5.json | cut -c1-500
