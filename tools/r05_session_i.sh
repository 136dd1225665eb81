#!/bin/bash
# round 5, GPU session I: the restored checkout rebuilt from scratch (source hash a9d0ab4b...): the whole -m gpu suite, the smoke, the driver's command
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -2 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 400 $O/bench_driver_cmd.json
