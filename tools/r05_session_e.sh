#!/bin/bash
# round 5, GPU session E: the audit build (tools/audit.py, tests/test_gpu_audit.py), every mutant's audit build, the driver's command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05e; mkdir -p $O gpurun_out/mutants; rm -f gpurun_out/mutants/audit.json
GCFR_HIP_LIB=$GRAFT_REPO_ROOT/geomconsistentfr_amd/lib/audit.so timeout 1200 python tools/audit.py --random 800 --family-seeds 8 --out $O/audit_product.json > $O/audit_product.log 2>&1; tail -c 1500 $O/audit_product.log
timeout 900 python -m pytest tests/test_gpu_audit.py -m gpu -q -p no:cacheprovider > $O/pytest_audit.log 2>&1; tail -3 $O/pytest_audit.log | cut -c1-300
timeout 3000 python tools/mutants.py run-audit > $O/mutants_audit.log 2>&1; cut -c1-330 $O/mutants_audit.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.json
