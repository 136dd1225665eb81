#!/bin/bash
# round 5, GPU session P: NORMALS_KERNEL_MIN_LIGHTS = 16 (host only): the tests that touch the render block's host side, config 5 stand-alone
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05p; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_normals.py tests/test_gpu_prepass_split.py tests/test_gpu_relightnet.py tests/test_gpu_configs.py -m gpu -q -p no:cacheprovider > $O/pytest_host_side.log 2>&1; grep -n "passed\|failed" $O/pytest_host_side.log | tail -2
python bench.py --no-cpu-baseline --no-worst-case --no-train-leg --size 512 --lights 18 --samples 320 --faces 1 --steps 300 > $O/bench_config5.json 2>/dev/null
python -c "
import json; d=json.loads(open('$O/bench_config5.json').read().strip().splitlines()[-1]); print('bench_config5', d['value'], d['ms_per_step'])"
