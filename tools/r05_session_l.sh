#!/bin/bash
# round 5, GPU session L (library 42970108..., PMC summary of session K): the -m gpu suite with the size-class test on even sizes, the
# smoke, the driver's command; the N > 1 launch rehearsed on the 1-GPU box (two ranks sharing the GPU over gloo -- RCCL refuses two
# ranks on one device), both workloads
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05l; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_full.log 2>&1; grep -n "passed\|failed" $O/pytest_full.log | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc $?"; tail -1 $O/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err; tail -c 300 $O/bench_driver_cmd.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --oversubscribe --steps 200 --warmup 50 --no-cpu-baseline --no-worst-case --no-train-leg > $O/bench_gpus2_oversubscribed.json 2> $O/bench_gpus2.err; tail -c 200 $O/bench_gpus2_oversubscribed.json; echo
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --oversubscribe --workload train --steps 8 --warmup 4 > $O/bench_train_gpus2_oversubscribed.json 2> $O/bench_train_gpus2.err; tail -c 200 $O/bench_train_gpus2_oversubscribed.json; echo
python bench.py --workload train --steps 20 > $O/bench_train.json 2>/dev/null; tail -c 200 $O/bench_train.json
