#!/usr/bin/env python3
"""Executed-work census of the march kernel (GPU box).

Loads the COUNTING build of the library (tools/build_variant.sh counters -DGCFR_COUNTERS ->
geomconsistentfr_amd/lib/counters.so; the product build compiles the counters out), runs the bench workload once and
reads the wave-level tallies the kernel adds to gcfr_options.counters: tiles, nominal / visited / bound-tested /
executed sample groups, lanes with an unmasked sample in an executed body, early exits, tie re-marches.  Together
with the PMC pass (tools/prof.sh -> SQ_INSTS_VALU) this gives VALU instructions per EXECUTED ray-step -- the figure
the nominal ray-step rate hides (DESIGN.md section 4.1 "Roofline").

usage: GCFR_HIP_LIB=geomconsistentfr_amd/lib/counters.so python tools/count_work.py [--faces 8] [--mask ellipse]
       [--depth-noise 0] [--size 256 --lights 1 --samples 160] [--tune knob=value,...] [--out profiles/x.json]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import RenderParams, _lib  # noqa: E402
from geomconsistentfr_amd import block as R  # noqa: E402


def knobs_tile_w(tune, size):
    for kv in tune.split(","):
        if kv.startswith("tile_w=") and int(kv[7:]):
            return int(kv[7:])
    return 16                                # the library's auto rule with the depth-bound skip on


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=8)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--lights", type=int, default=1)
    ap.add_argument("--samples", type=int, default=160)
    ap.add_argument("--mask", choices=["ellipse", "ones"], default="ellipse")
    ap.add_argument("--depth-noise", type=float, default=0.0)
    ap.add_argument("--argmin", action="store_true", help="the training-time (argmin) kernel variant")
    ap.add_argument("--tune", type=str, default="")
    ap.add_argument("--normals-in", action="store_true", help="normals as an input (rounds 1-3's step) instead of from depth")
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    L_ = _lib.load()
    ver = L_.gcfr_version().decode()
    if "+counters" not in ver:
        raise SystemExit("this is the product build (%s): run with GCFR_HIP_LIB=<counting build>, see the docstring" % ver)
    dev = torch.device("cuda:0")
    B, S, L, N = a.faces, a.size, a.lights, a.samples
    headline = (S == 256 and L == 1 and N == 160 and a.mask == "ellipse")
    if headline:
        prm = RenderParams()
        depth, mask, albedo, normals, light, amb = bench.synth_faces(B, 0)
    else:
        prm = RenderParams(n_samples=N, dt=0.8 / N)
        depth, mask, albedo, normals, light, amb = bench.synth_faces_sized(B, 0, S, L, a.mask)
    if a.depth_noise > 0:
        depth = depth + (a.depth_noise * np.random.default_rng(7).random(depth.shape)).astype(np.float32)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    tile_w = knobs_tile_w(a.tune, S)
    n_tiles = B * L * ((S + tile_w - 1) // tile_w) * ((S + 64 // tile_w - 1) // (64 // tile_w))
    counters = torch.zeros(_lib.N_COUNTERS + 4 * n_tiles, dtype=torch.int64, device=dev)   # tallies + per-tile records
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    opt = _lib.options(**knobs, counters=counters.data_ptr())
    cam = (1570.0 * S / 256.0, 1570.0 * S / 256.0, S / 2.0, S / 2.0, 1610.0)
    R.render_fwd(t(depth), t(mask), t(light).reshape(B, L, 3), t(amb).reshape(B, L), t(normals) if a.normals_in else None, t(albedo), prm,
                 want_argmin=a.argmin, camera=None if a.normals_in else cam, options=opt)
    torch.cuda.synchronize()
    c = dict(zip(_lib.COUNTER_NAMES, counters[:_lib.N_COUNTERS].cpu().tolist()))
    group = knobs.get("group", 0) or 4
    nominal = B * L * S * S * N
    out = {"library": ver, "workload": {"faces": B, "size": S, "lights": L, "samples": N, "mask": a.mask,
                                        "depth_noise": a.depth_noise, "argmin": a.argmin, "knobs": knobs},
           "counters": c, "nominal_ray_steps": nominal,
           # (tiles in the tile function's rough variant count their samples one by one: rough_samples, wave level)
           "executed_ray_steps_wave_level": (c["bodies"] * group + c.get("rough_samples", 0)) * 64,
           "executed_fraction_of_nominal": (c["bodies"] * group + c.get("rough_samples", 0)) * 64 / nominal,
           # (round 4) per executed wave-sample: how often does ANY lane lower its running minimum -- what an f32 pre-filter in front
           # of the exact body could at best avoid is the rest
           "wave_samples_in_which_some_lane_takes": c.get("wave_samples_taken", 0) / max(c.get("wave_samples", 0), 1),
           "lanes_taking_per_executed_wave_sample": c.get("lane_takes", 0) / max(c.get("wave_samples", 0), 1),
           "useful_lane_samples": c["lane_samples"],
           "lane_utilisation_of_executed_bodies": c["lane_samples"] / max(c["bodies"] * group * 64, 1),
           "groups_visited_per_tile": c["groups_visited"] / max(c["tiles"], 1),
           "bound_tests_per_tile": c["bound_tests"] / max(c["tiles"], 1),
           "bodies_per_tile": c["bodies"] / max(c["tiles"], 1),
           "nominal_groups_per_tile": c["groups_nominal"] / max(c["tiles"], 1)}
    print(json.dumps(out, indent=1))
    if a.out:
        json.dump(out, open(os.path.join(ROOT, a.out), "w"), indent=1)


if __name__ == "__main__":
    main()
