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
    a = ap.parse_args()
    dev = "cuda:0"
    if a.miopen_find:
        torch.backends.cudnn.benchmark = True
    tr = Trainer(TrainConfig(), device=dev)
    if a.channels_last:
        tr.model.to(memory_format=torch.channels_last)
        tr.patchgan.to(memory_format=torch.channels_last)
    batch = synthetic_batch(a.batch, 0, device=dev)
    for j in range(a.warmup):
        tr.step(batch, 200, j, log=False)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for j in range(a.steps):
        tr.step(batch, 200, j, log=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / a.steps
    print(json.dumps({"workload": "train step, batch %d, 256x256x160" % a.batch, "ms_per_step": 1e3 * dt,
                      "faces_per_sec": a.batch / dt, "ray_steps_per_sec": a.batch * 256 * 256 * 160 / dt}))
    if a.profile:
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
