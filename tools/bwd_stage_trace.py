#!/usr/bin/env python3
"""Where does a wave of render_bwd_single_light_kernel spend its time?  (GPU box, debug build:
tools/build_variant.sh bwdtrace -DGCFR_BWD_TRACE; GCFR_HIP_LIB=.../bwdtrace.so python tools/bwd_stage_trace.py)
Per wave and tile the kernel stamps the 100 MHz clock at its stage boundaries; this prints the mean time per stage."""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import _lib  # noqa: E402
from geomconsistentfr_amd.block import render_from_depth  # noqa: E402

L_ = _lib.load()
fn = L_.gcfr_debug_set_bwd_trace
fn.argtypes = [ctypes.c_void_p]
dev = torch.device("cuda:0")
B = 32
per_image = min(256, (2048 + B - 1) // B)
buf = torch.zeros(B * per_image * 4 * 8, dtype=torch.int64, device=dev)
assert fn(buf.data_ptr()) == 0
depth, mask, albedo, _, light, amb = bench.synth_faces(B, 100)
depth = depth + (2.0 * np.random.default_rng(5).random(depth.shape)).astype(np.float32)
t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
leaves = [t(x).requires_grad_() for x in (depth[:, None], albedo, light, amb)]
K = torch.zeros(1, 3, 3, dtype=torch.float64)
K[:, 0, 0] = K[:, 1, 1] = 1570.0
K[:, 2, 2] = 1.0
K[:, 0, 2] = K[:, 1, 2] = 128.0
for it in range(3):
    for l in leaves:
        l.grad = None
    o = render_from_depth(leaves[0], leaves[1], leaves[2], leaves[3], K, 1610.0, t(mask))
    (o["rendered_images"].sum() + o["shadow_mask_weights"].sum()).backward()
    torch.cuda.synchronize()
rec = buf.cpu().numpy().reshape(-1, 8)
rec = rec[rec[:, 6] > 0]
names = ["(1) shading", "(2) march bwd", "(4a) stencil terms + publish", "barrier wait", "gather + atomics issue", "2nd barrier"]
tiles = rec[:, 6].sum()
print("waves %d, tiles/wave %.1f" % (len(rec), rec[:, 6].mean()))
for i, n in enumerate(names):
    print("  %-30s %.2f us per tile (mean)" % (n, rec[:, i].sum() / tiles * 0.01))
print("  total per tile %.2f us; kernel span %.1f us" % (rec[:, :6].sum() / tiles * 0.01, (rec[:, 7].max() - rec[:, 7].min()) * 0.01))
