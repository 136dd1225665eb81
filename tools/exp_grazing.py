#!/usr/bin/env python3
"""Experiment (GPU box): march time vs light elevation, depth-bound skip on / off (gcfr_options.depth_bound_skip)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import RenderParams, _lib  # noqa: E402
from geomconsistentfr_amd import block as R  # noqa: E402

dev = torch.device("cuda:0")
B = 32
depth, mask, albedo, normals, light, amb = bench.synth_faces(B, 0)
prm = RenderParams()
tens = lambda a: torch.from_numpy(a).to(dev)
d_depth, d_mask, d_alb, d_nrm = tens(depth), R.mask_to_u8(tens(mask)).reshape(-1, 256, 256), tens(albedo), tens(normals)
d_amb = tens(amb).reshape(B, 1)
for lz in (0.9, 0.5, 0.2, 0.05, 0.0):
    l = np.array([[np.sqrt(max(1 - lz * lz, 0.0)) * 0.8, np.sqrt(max(1 - lz * lz, 0.0)) * 0.6, lz]] * B, np.float32)
    d_light = tens(l).reshape(B, 1, 3)
    row = []
    for zb in (0, 1):
        plan = R.RenderFwdPlan(B, 1, 256, 256, prm, dev, want_argmin=True, mask_batch=d_mask.shape[0],
                               options=_lib.options(depth_bound_skip=zb))
        for _ in range(5):
            plan(d_depth, d_mask, d_light, d_amb, d_nrm, d_alb)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            plan(d_depth, d_mask, d_light, d_amb, d_nrm, d_alb)
        e1.record()
        torch.cuda.synchronize()
        row.append(e0.elapsed_time(e1) / 50)
    print("light z %.2f: step %.3f ms without bounds, %.3f ms with  (x%.2f)" % (lz, row[0], row[1], row[0] / row[1]))
