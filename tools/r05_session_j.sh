#!/bin/bash
# round 5, GPU session J (final library, hash a9d0ab4b...): the N > 1 launch rehearsed on the 1-GPU box (two ranks sharing the GPU over
# gloo: launcher, barrier, max-over-ranks; RCCL refuses two ranks on one device), both workloads; soaks with fresh seeds
cd $GRAFT_REPO_ROOT; O=gpurun_out/r05j; mkdir -p $O
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --oversubscribe --steps 200 --warmup 50 --no-cpu-baseline --no-worst-case --no-train-leg > $O/bench_gpus2_oversubscribed.json 2> $O/bench_gpus2.err; tail -c 300 $O/bench_gpus2_oversubscribed.json
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29612 bench.py --gpus 2 --oversubscribe --workload train --steps 8 --warmup 4 > $O/bench_train_gpus2_oversubscribed.json 2> $O/bench_train_gpus2.err; tail -c 300 $O/bench_train_gpus2_oversubscribed.json
timeout 600 python tools/soak_parity.py --cases 10000 --seed 91 > $O/soak_argmin.json 2> $O/soak.err
timeout 600 python tools/soak_parity.py --cases 10000 --seed 92 --no-argmin > $O/soak_noargmin.json 2>> $O/soak.err
timeout 900 python tools/soak_parity.py --config5 64 --seed 94 > $O/soak_config5_argmin.json 2>> $O/soak.err
timeout 900 python tools/soak_backward.py --cases 3000 --seed 96 > $O/soak_backward.json 2>> $O/soak.err
cat $O/soak_*.json | cut -c1-260
