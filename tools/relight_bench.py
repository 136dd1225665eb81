#!/usr/bin/env python3
"""End-to-end relight rate beyond the bench line's leg (GPU box): inference.RelightSession at several batch sizes, with MIOpen's
immediate-mode picks (what bench.py's `relight_e2e` leg captures: no search, the leg must cost seconds) and with its searched
solvers (`miopen_find=True`: construction time included in the print).  One JSON line per configuration.
usage: tools/relight_bench.py [--faces 8,32] [--lights 1,11] [--iters 50] [--find 0|1]"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import scenes  # noqa: E402
from geomconsistentfr_amd import inference as inf  # noqa: E402
from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", default="8,32")
    ap.add_argument("--lights", default="1,11")
    ap.add_argument("--iters", type=int, default=50)
    ap.add_argument("--fold-bn", type=int, default=0, help="1: the network with its BatchNorms folded into the convolutions")
    ap.add_argument("--in-flight", default="2,4", help="also: this many sessions replayed round-robin on their own streams")
    ap.add_argument("--find", default="0,1", help="miopen_find settings, in this order (PyTorch caches a convolution's solver per process "
                                                  "and shape: whichever runs first decides for both -- use one process per setting)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    sd = {k: torch.from_numpy(v) for k, v in np.load(os.path.join(ROOT, "tests", "golden", "slt_checkpoint_epoch106.npz")).items()}
    net = RelightNetLightingTransfer()
    net.load_state_dict(sd, strict=True)
    net = net.float().to(dev).eval()
    if a.fold_bn:
        net = inf.fold_batchnorm(net)            # eval-mode BatchNorm folded into the convolutions (inference.fold_batchnorm)
    lights = torch.from_numpy(scenes.LIGHTS18[:11].copy()).to(dev)
    for B in [int(v) for v in a.faces.split(",")]:
        depth, mask, albedo, _n, _l, _a = scenes.synth_faces(B, 0)
        shade = 0.45 + 0.55 * np.clip(depth / 80.0, 0, 1)
        x = torch.from_numpy((albedo * shade[:, None]).transpose(0, 2, 3, 1).astype(np.float32).copy()).to(dev)
        m_u8 = torch.from_numpy((mask[0] * 255).astype(np.uint8)).to(dev)
        for L in [int(v) for v in a.lights.split(",")]:
            for find in [bool(int(v)) for v in a.find.split(",")]:
                t0 = time.perf_counter()
                sess = inf.RelightSession(net, B, m_u8, lights[:L], 0.5, device=dev, miopen_find=find)
                torch.cuda.synchronize()
                build_s = time.perf_counter() - t0
                for _ in range(5):
                    sess.run(x)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(a.iters):
                    sess.run(x)
                torch.cuda.synchronize()
                t = (time.perf_counter() - t0) / a.iters
                print(json.dumps({"faces": B, "lights": L, "miopen_find": find, "fold_bn": bool(a.fold_bn), "ms_per_pass": 1e3 * t,
                                  "images_per_sec": B * L / t, "faces_per_sec": B / t, "session_build_s": build_s}), flush=True)
                del sess
                # several sessions in flight on their own streams (independent batches, as bench.py's headline keeps four render
                # batches in flight): a batch-8 network pass does not fill the chip
                for n_fly in [int(v) for v in a.in_flight.split(",") if int(v) > 1]:
                    streams = [torch.cuda.Stream(device=dev) for _ in range(n_fly)]
                    many = []
                    for st in streams:
                        with torch.cuda.stream(st):
                            many.append(inf.RelightSession(net, B, m_u8, lights[:L], 0.5, device=dev))
                    torch.cuda.synchronize()

                    def go(n):
                        for i in range(n):
                            with torch.cuda.stream(streams[i % n_fly]):
                                many[i % n_fly].run(x)
                    go(2 * n_fly)
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    go(a.iters)
                    torch.cuda.synchronize()
                    t = (time.perf_counter() - t0) / a.iters
                    print(json.dumps({"faces": B, "lights": L, "miopen_find": find, "fold_bn": bool(a.fold_bn), "sessions_in_flight": n_fly, "ms_per_pass": 1e3 * t,
                                      "images_per_sec": B * L / t, "faces_per_sec": B / t}), flush=True)
                    del many


if __name__ == "__main__":
    main()
