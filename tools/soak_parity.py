#!/usr/bin/env python3
"""Randomised parity soak (GPU box): many random (depth, mask, light) cases, HIP workspace kernel vs the C
oracle.  Reports the worst min-distance error over unmasked pixels and the argmin agreement.  Evidence for
the 'decision parity' claims (magic-number rint, quad texels, skip, bounding-box pruning, k-split)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle  # noqa: E402
from geomconsistentfr_amd import RenderParams, light_prep, shadow_min_distance  # noqa: E402


def random_case(rng, H, W):
    r, c = np.mgrid[0:H, 0:W]
    kind = rng.integers(0, 6)
    depth = (0.3 * H * np.exp(-(((c - rng.uniform(0.3, 0.7) * W) / (0.25 * W)) ** 2
                                + ((r - rng.uniform(0.3, 0.7) * H) / (0.3 * H)) ** 2))).astype(np.float32)
    if kind == 0:
        depth += rng.random((H, W), dtype=np.float32)
    elif kind == 1:
        depth += (3 * np.sin(c / rng.uniform(3, 9)) * np.cos(r / rng.uniform(3, 9))).astype(np.float32)
    elif kind == 2:
        depth = np.round(depth)                       # plateaus -> exact ties between samples
    elif kind == 4:                                   # rough: rougher than the rays rise -- the tiles give the depth bounds up
        depth += (rng.uniform(20.0, 400.0) * rng.random((H, W))).astype(np.float32)      # (round 4: the rough loop)
    elif kind == 5:                                   # half smooth, half rough: both variants of a kernel in one launch
        depth += (rng.uniform(50.0, 400.0) * rng.random((H, W)) * (c > rng.uniform(0.3, 0.7) * W)).astype(np.float32)
    shift = rng.integers(0, 6)                        # depth sign / offset variants (exercise the depth-bound skip)
    if shift == 1:
        depth = -depth
    elif shift == 2:
        depth = depth - np.float32(0.15 * H)          # straddles zero
    elif shift == 3:
        depth = depth + np.float32(rng.choice([-1000.0, 1000.0]))
    mk = rng.integers(0, 4)
    if mk == 0:
        mask = rng.random((H, W)) > rng.uniform(0.05, 0.6)
    elif mk == 1:
        mask = (((c - rng.uniform(0.3, 0.7) * W) / (rng.uniform(0.1, 0.45) * W)) ** 2
                + ((r - rng.uniform(0.3, 0.7) * H) / (rng.uniform(0.1, 0.45) * H)) ** 2) < 1
    elif mk == 2:
        r0, c0 = rng.integers(0, H - 4), rng.integers(0, W - 4)
        mask = (r >= r0) & (r < r0 + rng.integers(1, H - r0)) & (c >= c0) & (c < c0 + rng.integers(1, W - c0))
    else:
        mask = np.ones((H, W), bool)
    l = rng.standard_normal(3)
    if rng.random() < 0.25:
        l[:2] *= 0.01                                 # light nearly on the optical axis (mid/mid branch)
    if rng.random() < 0.15:
        l[rng.integers(0, 2)] = 0.0                   # axis-aligned rays: the -1 wrap column / row
    return depth, mask.astype(np.uint8), l.astype(np.float32)


def run_soak(n_cases, seed=0, options=None, sizes=None, want_argmin=True, pixels_mask=False):
    """`n_cases` random (depth, mask, light) cases in batches of 8: HIP workspace kernel vs the C oracle.
    Returns the tallies (pixels compared, lit/masked disagreements, worst min-distance error, argmin differences)."""
    rng = np.random.default_rng(seed)
    rng_b = np.random.default_rng([seed, 0xB0DE])        # its own stream: the cases of earlier rounds' seeds stay what they were
    dev = torch.device("cuda:0")
    sizes = sizes or [(64, 64, 48), (96, 128, 80), (130, 70, 37), (128, 128, 160), (256, 256, 160)]
    n_boundary = 0
    worst = {"abs": 0.0, "rel": 0.0}
    n_pix = n_arg_diff = n_lit_mismatch = 0
    t0 = time.time()
    B = 8
    for it in range(n_cases // B):
        H, W, N = sizes[it % len(sizes)]
        cases = [random_case(rng, H, W) for _ in range(B)]
        depth = np.stack([c[0] for c in cases])
        mask = np.stack([c[1] for c in cases])
        lights = np.stack([c[2] for c in cases])
        # per batch: depth scale (steep / flat surfaces: the plane-fit bounds clamp their slopes at 4) and light
        # distance (near lights: rays far from parallel, small |BC|)
        scale = [1.0, 1.0, 0.01, 12.0, 300.0][it % 5] if it % 3 == 0 else 1.0
        depth = (depth * np.float32(scale)).astype(np.float32)
        ld = [4013.0, 4013.0, 60.0, 500.0, 1.0e5, 30.0][(it // 5) % 6]
        prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N, light_distance=ld)
        _, pt = light_prep(torch.from_numpy(lights).to(dev), prm)
        _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0, light_distance=ld)
        assert np.array_equal(pt.cpu().numpy(), pt_o)
        # round 6: one case in six gets its light POINT moved exactly onto -- or one f32 ulp either side of -- a boundary of the
        # reference's nine-way end-point branch (T8:386-431: x = -W/2, W/2 - 1; y = 1 - H/2, H/2), where `<` / `<=` decide which of
        # the nine forms every ray of the image takes; both sides are handed the same point
        for b in range(B):
            if rng_b.random() < 1.0 / 6.0:
                axis = int(rng_b.integers(0, 2))
                edge = [(-(W / 2.0), W - W / 2.0 - 1), (1 - H / 2.0, H / 2.0)][axis][int(rng_b.integers(0, 2))]
                v = np.float32(edge)
                step = int(rng_b.integers(-1, 2))
                if step:
                    v = np.nextafter(v, np.float32(np.inf * step))
                pt_o[b, axis] = v
                n_boundary += 1
        pt = torch.from_numpy(pt_o).to(dev)
        md, am = shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev),
                                     pt.reshape(B, 1, 3), prm, options=options, want_argmin=want_argmin)
        md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(0.025, 0.8 / N, N))
        if pixels_mask:    # gcfr_options.pixels = 1: a pixel whose own mask cell is zero carries the masked value, every other pixel the oracle's
            md_o[:, 0][mask == 0] = 1e6
            am_o[:, 0][mask == 0] = -1
        md, am = md.cpu().numpy(), (am.cpu().numpy() if am is not None else am_o)   # (kernels without argmin: distances only)
        lit_o, lit = md_o < 1e5, md < 1e5
        n_lit_mismatch += int((lit_o != lit).sum())
        if pixels_mask:
            n_lit_mismatch += int((md[~lit] != md_o[~lit]).sum()) + (int((am[~lit] != -1).sum()) if want_argmin else 0)
        both = lit & lit_o
        err = np.abs(md[both] - md_o[both])
        if err.size:
            worst["abs"] = max(worst["abs"], float(err.max()))
            worst["rel"] = max(worst["rel"], float((err / np.maximum(np.abs(md_o[both]), 1.0)).max()))
        n_pix += int(both.sum())
        n_arg_diff += int((am[both] != am_o[both]).sum())
    return {"cases": (n_cases // B) * B, "seed": seed, "cases_with_the_light_point_on_a_branch_boundary": n_boundary,
            "pixels_compared": n_pix, "lit_mask_mismatches": n_lit_mismatch,
            "max_abs_err_min_dist": worst["abs"], "max_rel_err_min_dist": worst["rel"],
            "argmin_differences": n_arg_diff, "argmin_difference_rate": n_arg_diff / max(n_pix, 1),
            "seconds": time.time() - t0}


def run_config5(n_faces, seed=0, want_argmin=True):
    """The soak's leg at BASELINE configs[4]'s shape: `n_faces` random 512 x 512 faces, 18 random lights each
    (all lights of a face in one launch, as the product shards them), 320 samples, the grid schedule forced (ksplit = 0: bounds
    skip, trailing loop, horizon tables).  Bit equality of min_dist and argmin against the C oracle."""
    from geomconsistentfr_amd import _lib
    rng = np.random.default_rng(seed)
    dev = torch.device("cuda:0")
    H = W = 512
    N, L = 320, 18
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
    tt = c_oracle.sample_table(0.025, 0.8 / N, N)
    n_pix = n_md_diff = n_arg_diff = 0
    t0 = time.time()
    for f in range(n_faces):
        depth, mask, _ = random_case(rng, H, W)
        lights = rng.standard_normal((L, 3)).astype(np.float32)
        lights[:, 2] = np.abs(lights[:, 2]) * rng.choice([1.0, 0.1])
        _, pt = light_prep(torch.from_numpy(lights).to(dev), prm)
        md, am = shadow_min_distance(torch.from_numpy(depth[None]).to(dev), torch.from_numpy(mask[None]).to(dev), pt.reshape(1, L, 3), prm,
                                     options=_lib.options(ksplit=0), want_argmin=want_argmin)
        _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
        md_o, am_o = c_oracle.shadow_min_distance(depth[None], mask[None], pt_o[None], tt)
        md = md.cpu().numpy()
        n_pix += md.size
        n_md_diff += int((md.view(np.int32) != md_o.view(np.int32)).sum())
        if want_argmin:
            lit = md_o < 1e5
            n_arg_diff += int((am.cpu().numpy()[lit] != am_o[lit]).sum())
    return {"leg": "config 5 shape: 512 x 512 x 320, 18 lights per face, grid schedule forced", "faces": n_faces, "seed": seed,
            "pixels_compared": n_pix, "min_dist_bit_differences": n_md_diff, "argmin_differences": n_arg_diff, "argmin_kernel": want_argmin,
            "seconds": time.time() - t0}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config5", type=int, default=0, help="N faces at configs[4]'s shape (512 x 512 x 320, 18 lights) instead of the mixed-size soak")
    ap.add_argument("--cases", type=int, default=240)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--tune", type=str, default="", help="comma list of gcfr_options knobs, e.g. ksplit=0,tile_w=16")
    ap.add_argument("--no-argmin", action="store_true", help="the inference kernels (six waves / SIMD, no argmin output)")
    a = ap.parse_args()
    from geomconsistentfr_amd import _lib
    if a.config5:
        r = run_config5(a.config5, a.seed, want_argmin=not a.no_argmin)
        r.update(library=_lib.load().gcfr_version().decode())
        print(json.dumps(r))
        return
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    r = run_soak(a.cases, a.seed, _lib.options(**knobs) if knobs else None, want_argmin=not a.no_argmin,
                 pixels_mask=knobs.get("pixels", 0) == 1)
    r.update(library=_lib.load().gcfr_version().decode(), knobs=knobs, argmin_kernel=not a.no_argmin)
    print(json.dumps(r))


if __name__ == "__main__":
    main()
