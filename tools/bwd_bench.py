#!/usr/bin/env python3
"""Render-block forward + backward at BASELINE configs[2]'s batch (B = 32 faces, 256 x 256 x 160), outside the
network: render_from_depth (prepass, march with fused normals + shading) and its fused one-launch backward
(gcfr_render_bwd) + light-prep backward, `--iters` times.  Used under rocprofv3 for the backward kernels' profiles
(tools/prof.sh <tag> bwd) and stand-alone for event timings.  Depth = synthetic faces + uniform noise of
`--depth-noise` (an untrained network's depth is rough: that is the training-time march)."""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd.block import render_from_depth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=32)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--depth-noise", type=float, default=2.0)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    B = a.faces
    depth, mask, albedo, _, light, amb = bench.synth_faces(B, 100)
    depth = depth + (a.depth_noise * np.random.default_rng(5).random(depth.shape)).astype(np.float32)
    rng = np.random.default_rng(1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    leaves = [t(x).requires_grad_() for x in (depth[:, None], albedo, light, amb)]
    d_mask = t(mask)
    G_r, G_w = t(rng.random((B, 3, 256, 256), dtype=np.float32)), t(rng.random((B, 256, 256), dtype=np.float32))
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 1570.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    t_f, t_b = [], []
    for it in range(a.iters + 3):
        for l in leaves:
            l.grad = None
        ev[0].record()
        o = render_from_depth(leaves[0], leaves[1], leaves[2], leaves[3], K, 1610.0, d_mask)
        loss = (o["rendered_images"] * G_r).sum() + (o["shadow_mask_weights"] * G_w).sum()
        ev[1].record()
        loss.backward()
        ev[2].record()
        torch.cuda.synchronize()
        if it >= 3:
            t_f.append(ev[0].elapsed_time(ev[1]))
            t_b.append(ev[1].elapsed_time(ev[2]))
    print(json.dumps({"faces": B, "depth_noise": a.depth_noise, "forward_ms_incl_loss": float(np.median(t_f)),
                      "backward_ms_incl_loss_backward": float(np.median(t_b))}))


if __name__ == "__main__":
    main()
