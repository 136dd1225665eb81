#!/usr/bin/env python3
"""Work census of the TRAINING march (GPU box, counting build): batch 32, argmin variant, normals fused, depth / light of a
freshly initialised RelightNet on the synthetic training batch (bench.py's `train_depth` data) or the smooth bench faces,
with RenderParams.pixels = "all" and "mask" -- where the march's samples are and what leaving out the pixels outside the mask
removes.  usage: GCFR_HIP_LIB=geomconsistentfr_amd/lib/counters.so python tools/count_train_march.py [--data train_depth|synthetic]"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import _lib  # noqa: E402


class _Rk:
    rank, world, dist = 0, 1, None
    dev = torch.device("cuda:0")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--data", default="train_depth")
    ap.add_argument("--faces", type=int, default=32)
    a = ap.parse_args()
    assert "+counters" in _lib.load().gcfr_version().decode(), "run with GCFR_HIP_LIB=<counting build>"
    out = {}
    for px in ("all", "mask"):
        rig = bench.RenderRig(_Rk(), a.faces, data=a.data, from_depth=True, want_argmin=True, streams=1, graph=False, pixels=px)
        n_tiles = a.faces * 16 * 64
        counters = torch.zeros(_lib.N_COUNTERS + 4 * n_tiles, dtype=torch.int64, device=_Rk.dev)
        rig.plans[0].options = _lib.options(counters=counters.data_ptr(), pixels=int(px == "mask"))
        rig.plans[0](*rig.inputs[0])
        torch.cuda.synchronize()
        c = dict(zip(_lib.COUNTER_NAMES, counters[:_lib.N_COUNTERS].cpu().tolist()))
        mask = rig.batches[0][1]
        c["mask_fraction"] = float((mask != 0).float().mean())
        c["executed_wave_samples"] = c["bodies"] * 4 + c["rough_samples"]
        out[px] = c
    out["executed_ratio_mask_over_all"] = out["mask"]["executed_wave_samples"] / max(out["all"]["executed_wave_samples"], 1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
