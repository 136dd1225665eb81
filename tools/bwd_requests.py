#!/usr/bin/env python3
"""Atomic REQUESTS of the fused backward by source (GPU; counting build: tools/build_variant.sh bwd_count -DGCFR_FAST_BUILD
-DGCFR_BWD_COUNT, run with GCFR_HIP_LIB=geomconsistentfr_amd/lib/bwd_count.so).

render_bwd_single_light_kernel adds its gradients to grad_depth with global f32 atomics from four places: the pixel's own cell
(stencil gather + own-depth terms, one per pixel), the flush of the LDS corner window (one lane per column, row by row), the
fallback when a tile's corners do not fit the window (run-merged, direct), and what the normals stencil owes to pixels outside
the tile / on the image border.  Per source: wave instructions, 64-B lines touched (each a read-modify-write at the memory
side: the REQUESTS), elements.  One launch, B = 32 x 256 x 256, the dense upstream gradient of tools/bwd_bench.py and the
masked one of the training step (the losses multiply by the mask).  -> profiles/r05_bwd_requests.json"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import _lib  # noqa: E402
from geomconsistentfr_amd.block import render_from_depth  # noqa: E402

SOURCES = ("own pixel (gcfr_backward.hip: stencil gather + own-depth terms)", "corner window flush", "fallback corners (run-merged, direct)",
           "stencil halo / image border (normals_bwd_scatter)")


def main():
    dev = torch.device("cuda:0")
    L = _lib.load()
    assert b"gcfr" in L.gcfr_version()
    try:
        setter = L.gcfr_debug_set_bwd_count
    except AttributeError:
        sys.exit("this library is not a -DGCFR_BWD_COUNT build (GCFR_HIP_LIB=.../bwd_count.so)")
    setter.argtypes = [ctypes.c_void_p]
    B = 32
    depth, mask, albedo, _, light, amb = bench.synth_faces(B, 100)
    depth = depth + (2.0 * np.random.default_rng(5).random(depth.shape)).astype(np.float32)
    rng = np.random.default_rng(1)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    d_mask = t(mask)
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 1570.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    G_r, G_w = t(rng.random((B, 3, 256, 256), dtype=np.float32)), t(rng.random((B, 256, 256), dtype=np.float32))
    out = {}
    for name, m in (("dense upstream gradient (tools/bwd_bench.py)", None), ("masked upstream gradient (the training step's)", d_mask.float())):
        leaves = [t(x).requires_grad_() for x in (depth[:, None], albedo, light, amb)]
        o = render_from_depth(leaves[0], leaves[1], leaves[2], leaves[3], K, 1610.0, d_mask)
        gr, gw = (G_r, G_w) if m is None else (G_r * m[:, None], G_w * m)
        loss = (o["rendered_images"] * gr).sum() + (o["shadow_mask_weights"] * gw).sum()
        cnt = torch.zeros(12, dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        assert setter(cnt.data_ptr()) == 0
        loss.backward()
        torch.cuda.synchronize()
        assert setter(None) == 0
        c = cnt.cpu().numpy().reshape(4, 3)
        rows = {SOURCES[i]: {"wave_instructions": int(c[i, 0]), "lines_64B": int(c[i, 1]), "elements": int(c[i, 2]),
                             "lines_per_instruction": float(c[i, 1]) / max(int(c[i, 0]), 1)} for i in range(4)}
        lines = int(c[:, 1].sum())
        out[name] = {"by_source": rows, "lines_total": lines, "elements_total": int(c[:, 2].sum()),
                     "read_modify_write_MB_at_128B_per_line": lines * 128 / 1e6,
                     "pixels": B * 65536, "lines_per_pixel": lines / (B * 65536.0)}
        print("==", name)
        for k, v in rows.items():
            print("   %-72s %9d instr %10d lines %10d elements  (%.2f lines / instr)" % (k, v["wave_instructions"], v["lines_64B"], v["elements"], v["lines_per_instruction"]))
        print("   total %d lines = %.1f MB of read-modify-write traffic at 128 B per line; %.3f lines per pixel" % (lines, lines * 128 / 1e6, lines / (B * 65536.0)))
    path = os.path.join(ROOT, "gpurun_out", "r05_bwd_requests.json")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
