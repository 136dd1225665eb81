#!/usr/bin/env python3
"""Timing of the full training step (BASELINE configs[2]: batch 32, forward+backward inside the step) on
one GPU, with a per-phase breakdown from torch.profiler.  Not the headline bench (bench.py)."""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--miopen-find", action="store_true", help="torch.backends.cudnn.benchmark = True")
    ap.add_argument("--channels-last", action="store_true")
    ap.add_argument("--grep", type=str, default="", help="with --profile: only rows whose name contains this")
    ap.add_argument("--ddp", action="store_true", help="wrap in DistributedDataParallel over RCCL (launch with torchrun)")
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = "cuda:%d" % local_rank
    torch.cuda.set_device(local_rank)
    if a.ddp:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
    if a.miopen_find:
        torch.backends.cudnn.benchmark = True
    tr = Trainer(TrainConfig(miopen_find=a.miopen_find), device=dev, distributed=a.ddp)
    if a.channels_last:
        tr.model.to(memory_format=torch.channels_last)
        tr.patchgan.to(memory_format=torch.channels_last)
    batch = synthetic_batch(a.batch, rank * 1_000_000, device=dev)      # whole faces per rank, seed = rank*1e6 + index
    for j in range(a.warmup):
        tr.step(batch, 200, j, log=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for j in range(a.steps):
        tr.step(batch, 200, j, log=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.steps
    if rank == 0:
        print(json.dumps({"workload": "train step, batch %d per GPU x %d GPU(s), 256x256x160" % (a.batch, world),
                          "ddp": a.ddp, "ms_per_step": 1e3 * dt, "faces_per_sec": world * a.batch / dt,
                          "ray_steps_per_sec": world * a.batch * 256 * 256 * 160 / dt}))
    if a.ddp:
        dist.barrier()
    if a.profile and rank == 0:
        from torch.profiler import ProfilerActivity, profile
        with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
            for j in range(3):
                tr.step(batch, 200, j + 1, log=False)
            torch.cuda.synchronize()
        if a.grep:
            for e in prof.key_averages():
                if a.grep in e.key:
                    print("%-90s calls %4d  device total %9.1f us  avg %8.1f us" % (e.key[:90], e.count, e.device_time_total, e.device_time_total / max(e.count, 1)))
        else:
            print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=25, max_name_column_width=60))


if __name__ == "__main__":
    main()
    if torch.distributed.is_available() and torch.distributed.is_initialized():
        torch.distributed.destroy_process_group()
