#!/usr/bin/env python3
"""Randomised soak of the backward kernels (GPU box): for random even sizes, masks, lights and upstream gradients,
  (1) the fused single-light backward (gcfr_render_bwd, L = 1: render_from_depth's autograd) against the three-kernel path
      (gcfr_shade_bwd -> gcfr_shadow_bwd -> gcfr_normals_bwd), which autograd through the materialised oracle pins;
  (2) the multi-light fused backward (L = 2 ... 4 through the C ABI) against the sum of L single-light runs.
Depth / albedo gradients come from the same device functions on both sides and agree up to f32 atomic ordering (gate 5e-6 of the
gradient's maximum); light / ambient gradients are sums over the image of a stage the fused kernels evaluate in f32 and the
stand-alone kernel in f64 (gate 5e-5).
The advisor's round-2 finding (wrapped-column run key at W % 16 != 0) is the kind of bug this is for."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomconsistentfr_amd import RenderParams, _lib, render  # noqa: E402
from geomconsistentfr_amd import block as R  # noqa: E402
from geomconsistentfr_amd.normals import depth_to_normals  # noqa: E402


def camera(f, H, W, dev):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    return K.to(dev)


def random_inputs(rng, B, H, W):
    r, c = np.mgrid[0:H, 0:W]
    depth = (0.3 * H * np.exp(-(((c - rng.uniform(0.2, 0.8) * W) / (0.3 * W)) ** 2 + ((r - rng.uniform(0.2, 0.8) * H) / (0.3 * H)) ** 2)))
    depth = np.stack([depth + rng.uniform(0, 3) * rng.random((H, W)) for _ in range(B)]).astype(np.float32)
    kind = rng.integers(0, 3)
    mask = np.ones((B, H, W), np.uint8) if kind == 0 else (rng.random((B, H, W)) > rng.uniform(0.05, 0.5)).astype(np.uint8)
    albedo = rng.random((B, 3, H, W), dtype=np.float32)
    return depth, mask, albedo


def rel(a, b, floor=0.0):
    """max |a - b| relative to max |b| (+ `floor`: light / ambient gradients are sums over B*H*W pixels of O(1) terms that can
    cancel to almost nothing; their f32-vs-f64 stage difference is absolute, ~1e-8 per pixel, not relative to the sum)"""
    s = float(b.abs().max()) + floor
    return float((a - b).abs().max()) / max(s, 1e-12)


def oracle_soak(a):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import materialised as M
    from normals_restatement import depth_to_normals as oracle_normals
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    names = ("depth", "albedo", "light", "ambient")
    worst = dict.fromkeys(names, 0.0)
    over, t0 = [], time.time()
    for it in range(a.oracle):
        B = int(rng.integers(1, 3))
        H, W = 2 * int(rng.integers(8, 25)), 2 * int(rng.integers(8, 25))
        N = int(rng.integers(8, 41))
        depth, mask, albedo = random_inputs(rng, B, H, W)
        f, zoff = float(rng.uniform(300, 2000)), float(rng.uniform(100, 2000))
        light = rng.standard_normal((B, 3)).astype(np.float32)
        light[:, 2] = np.abs(light[:, 2]) + 0.05
        amb = (0.3 + 0.4 * rng.random(B)).astype(np.float32)
        G = {k: rng.standard_normal(s_).astype(np.float32) for k, s_ in
             [("shadow_mask_weights", (B, H, W)), ("final_shading", (B, H, W)), ("rendered_images", (B, 3, H, W))]}
        prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
        K = camera(f, H, W, "cpu")
        # oracle: autograd on the host
        cl = [torch.from_numpy(x).clone().requires_grad_() for x in (depth[:, None], albedo, light, amb)]
        n = oracle_normals(cl[0] + zoff, K)
        n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], 1)
        o = M.render_block(cl[0], cl[1], cl[2], cl[3], n, torch.from_numpy(mask), M.BlockParams(n_samples=N, t0=0.025, dt=0.8 / N))
        sum((o[k].reshape(g.shape) * torch.from_numpy(g)).sum() for k, g in G.items()).backward()
        # product: fused forward + fused backward
        gl = [t(x).requires_grad_() for x in (depth[:, None], albedo, light, amb)]
        oh = R.render_from_depth(gl[0], gl[1], gl[2], gl[3], K.to(dev), zoff, t(mask), prm)
        sum((oh[k] * t(g)).sum() for k, g in G.items()).backward()
        for name, c_, g_ in zip(names, cl, gl):
            e = rel(g_.grad.cpu().double(), c_.grad.double(), 2e-4 * B * H * W if name in ("light", "ambient") else 0.0)
            worst[name] = max(worst[name], e)
            # (the light-point gradient is a sum dominated by a few ill-conditioned pixels -- near-vertical rays when the light's
            #  projection falls next to the image box, slopes of 1e3 ... 1e4 -- which the reference's autograd evaluates in f32 and
            #  the kernels' chain rule in f64: on these 16 ... 48-pixel images the two differ by up to ~1 % there; on the 256 x 256
            #  golden batches they agree to 1e-4, tests/test_gpu_backward.py)
            if e > (2e-2 if name == "light" else 2e-4):
                over.append(dict(case=it, what=name, rel=e, B=B, H=H, W=W, N=N, product=g_.grad.cpu().numpy().tolist() if name in ("light", "ambient") else None,
                                 oracle=c_.grad.numpy().tolist() if name in ("light", "ambient") else None, light=light.tolist(),
                                 light_pt=oh["unit_light_direction"].detach().reshape(B, 3).mul(4013.0).cpu().numpy().tolist()))
    print(json.dumps({"mode": "fused single-light backward vs autograd through the materialised oracle", "cases": a.oracle, "seed": a.seed,
                      "worst_relative_difference": worst, "gates": {"depth, albedo, ambient": 2e-4, "light": 2e-2}, "cases_over_the_gate": over, "ok": len(over) == 0,
                      "seconds": time.time() - t0}))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=400)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--only", type=int, default=-1, help="replay one case of the sequence and print its details")
    ap.add_argument("--oracle", type=int, default=0,
                    help="instead of the kernel-vs-kernel soak: this many SMALL random cases (sizes 16 ... 48, <= 40 samples) of the fused "
                         "single-light backward against autograd through oracle/materialised.py + normals_restatement.py on the host "
                         "-- the torch port that is bit-equal to the reference in forward and matches its autograd gradients")
    a = ap.parse_args()
    if a.oracle > 0:
        return oracle_soak(a)
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(a.seed)
    L_ = _lib.load()
    worst = {"single_depth": 0.0, "single_albedo": 0.0, "single_light": 0.0, "single_ambient": 0.0, "multi_depth": 0.0, "multi_albedo": 0.0,
             "multi_light": 0.0}
    worst_case = {}
    t0 = time.time()
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    for it in range(a.cases):
        B = int(rng.integers(1, 4))
        H, W = 2 * int(rng.integers(8, 81)), 2 * int(rng.integers(8, 81))
        N = int(rng.integers(8, 97))
        prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
        depth, mask, albedo = random_inputs(rng, B, H, W)
        f, zoff = float(rng.uniform(300, 2000)), float(rng.uniform(100, 2000))
        K = camera(f, H, W, dev)
        # ---- (1) fused single-light vs three kernels
        light = rng.standard_normal((B, 3)).astype(np.float32)
        if rng.random() < 0.3:
            light[:, rng.integers(0, 2)] *= 30.0                   # light far outside the image on one axis: wrapped columns / rows
        light[:, 2] = np.abs(light[:, 2]) + 0.05
        amb = (0.3 + 0.4 * rng.random(B)).astype(np.float32)
        G = {k: t(rng.standard_normal(s).astype(np.float32)) for k, s in
             [("shadow_mask_weights", (B, H, W)), ("final_shading", (B, H, W)), ("rendered_images", (B, 3, H, W))]}
        run = a.only < 0 or it == a.only
        grads = []
        for fused in ((True, False) if run else ()):
            leaves = [t(x).requires_grad_() for x in (depth[:, None], albedo, light, amb)]
            if fused:
                o = R.render_from_depth(leaves[0], leaves[1], leaves[2], leaves[3], K, zoff, t(mask), prm)
            else:
                n = depth_to_normals(leaves[0], K, z_offset=zoff)
                o = render(leaves[0], leaves[1], leaves[2], leaves[3], n, t(mask), prm)
            sum((o[k] * g).sum() for k, g in G.items()).backward()
            grads.append([l.grad for l in leaves])
        for name, x, y in (zip(("single_depth", "single_albedo", "single_light", "single_ambient"), *grads) if run else ()):
            e = rel(x, y, 2e-4 * B * H * W if name in ("single_light", "single_ambient") else 0.0)
            if a.only >= 0:
                print(name, "rel", e, "fused", x.flatten()[:4].tolist(), "three-kernel", y.flatten()[:4].tolist())
            if e > worst[name]:
                worst[name], worst_case[name] = e, dict(case=it, B=B, H=H, W=W, N=N)
        # ---- (2) multi-light fused vs the sum of single-light runs
        L = int(rng.integers(2, 5))
        lights = rng.standard_normal((B, L, 3)).astype(np.float32)
        lights[..., 2] = np.abs(lights[..., 2]) + 0.05
        ambs = (0.3 + 0.4 * rng.random((B, L))).astype(np.float32)
        Gm = t(rng.standard_normal((B, L, 3, H, W)).astype(np.float32))
        cam = (f, f, W / 2.0, H / 2.0, zoff)
        if not run:
            continue
        d3, al, li, am, mk = t(depth), t(albedo), t(lights), t(ambs), t(mask)
        o = R.render_fwd(d3, mk, li, am, None, al, prm, want_argmin=True, camera=cam)
        g_alb, g_depth = torch.empty_like(al), torch.zeros_like(d3)
        g_pt = torch.zeros((B, L, 3), dtype=torch.float64, device=dev)
        g_amb = torch.zeros((B, L), dtype=torch.float64, device=dev)
        tt = R.sample_table(prm, dev)
        _lib.check(L_.gcfr_render_bwd(d3.data_ptr(), al.data_ptr(), o["light_pt"].data_ptr(), am.data_ptr(), o["minimum_distance"].data_ptr(),
                                      o["argmin"].data_ptr(), o["surface_normals"].data_ptr() if it % 2 else None, B, L, H, W, N, tt.data_ptr(),
                                      *cam[:4], cam[4], 1, 0.5, None, None, None, Gm.data_ptr(), None, g_alb.data_ptr(), g_depth.data_ptr(),
                                      g_pt.data_ptr(), g_amb.data_ptr(), torch.cuda.current_stream().cuda_stream), "gcfr_render_bwd")
        s_alb, s_depth, s_light = torch.zeros_like(al), torch.zeros_like(d3), []
        for l in range(L):
            dl, a_l = d3[:, None].clone().requires_grad_(), al.clone().requires_grad_()
            l_l, m_l = li[:, l].clone().requires_grad_(), am[:, l].clone().requires_grad_()
            r = R.render_from_depth(dl, a_l, l_l, m_l, K, zoff, mk, prm)
            (r["rendered_images"] * Gm[:, l]).sum().backward()
            s_alb += a_l.grad
            s_depth += dl.grad[:, 0]
            s_light.append(l_l.grad)
        # light gradient of the multi-light call: grad_light_pt -> through light prep, as the single path does
        gl = torch.empty((B * L, 3), dtype=torch.float32, device=dev)
        _lib.check(L_.gcfr_light_prep_bwd(li.reshape(-1, 3).contiguous().data_ptr(), B * L, 1, 0.0, float(prm.light_distance), None,
                                          g_pt.data_ptr(), gl.data_ptr(), torch.cuda.current_stream().cuda_stream), "gcfr_light_prep_bwd")
        for name, x, y in (("multi_depth", g_depth, s_depth), ("multi_albedo", g_alb, s_alb),
                           ("multi_light", gl.reshape(B, L, 3), torch.stack(s_light, 1))):
            e = rel(x, y, 2e-4 * B * H * W if name == "multi_light" else 0.0)
            if e > worst[name]:
                worst[name], worst_case[name] = e, dict(case=it, B=B, H=H, W=W, N=N, L=L)
            if a.only >= 0:
                d = (x - y).abs()
                idx = [int(v) for v in np.unravel_index(int(d.argmax()), d.shape)]
                print(name, "rel", e, "max|ref|", float(y.abs().max()), "at", idx, "got", float(x.flatten()[int(d.argmax())]),
                      "ref", float(y.flatten()[int(d.argmax())]), "n_diff>1e-5*max", int((d > 1e-5 * y.abs().max()).sum()))
        if a.only >= 0:   # a third opinion: the three-kernel path, light by light
            t_depth = torch.zeros_like(d3)
            for l in range(L):
                dl, a_l = d3[:, None].clone().requires_grad_(), al.clone().requires_grad_()
                l_l, m_l = li[:, l].clone().requires_grad_(), am[:, l].clone().requires_grad_()
                nrm = depth_to_normals(dl, K, z_offset=zoff)
                r = render(dl, a_l, l_l, m_l, nrm, mk, prm)
                (r["rendered_images"] * Gm[:, l]).sum().backward()
                t_depth += dl.grad[:, 0]
                dmu = (s_light[l] - l_l.grad).abs().max()
                print("light", l, "single-fused vs three-kernel light grad diff", float(dmu))
            print("multi vs three-kernel depth:", rel(g_depth, t_depth), " singles-sum vs three-kernel depth:", rel(s_depth, t_depth))
            for l in range(L):      # the multi-light kernel with ONE light's upstream gradient at a time, both normal sources
                Gl = torch.zeros_like(Gm)
                Gl[:, l] = Gm[:, l]
                dl, a_l = d3[:, None].clone().requires_grad_(), al.clone().requires_grad_()
                l_l, m_l = li[:, l].clone().requires_grad_(), am[:, l].clone().requires_grad_()
                r = R.render_from_depth(dl, a_l, l_l, m_l, K, zoff, mk, prm)
                (r["rendered_images"] * Gm[:, l]).sum().backward()
                for nf in (o["surface_normals"], None):
                    g_depth.zero_(), g_pt.zero_(), g_amb.zero_()
                    _lib.check(L_.gcfr_render_bwd(d3.data_ptr(), al.data_ptr(), o["light_pt"].data_ptr(), am.data_ptr(), o["minimum_distance"].data_ptr(),
                                                  o["argmin"].data_ptr(), None if nf is None else nf.data_ptr(), B, L, H, W, N, tt.data_ptr(),
                                                  *cam[:4], cam[4], 1, 0.5, None, None, None, Gl.data_ptr(), None, g_alb.data_ptr(), g_depth.data_ptr(),
                                                  g_pt.data_ptr(), g_amb.data_ptr(), torch.cuda.current_stream().cuda_stream), "gcfr_render_bwd")
                    d = (g_depth - dl.grad[:, 0]).abs()
                    bad = torch.nonzero(d > 1e-5 * dl.grad.abs().max()).cpu().numpy().tolist()
                    print("light", l, "normals", "read" if nf is not None else "recomputed", "rel", rel(g_depth, dl.grad[:, 0]), "bad pixels", bad[:8])
                    if bad and nf is None:     # a ring of eight = ONE pixel's stencil: look at that pixel's Lambert term in the forward
                        bb, rr, cc = bad[0][0], int(np.median([q[1] for q in bad])), int(np.median([q[2] for q in bad]))
                        fs, amv = o["full_shading"][bb, l, rr, cc], am[bb, l]
                        print("   centre pixel", (bb, rr, cc), "forward full_shading - ambient =", float(fs - amv), "(0.5 * max(n.l, 0))")
            for b_ in range(B):
                for l in range(L):
                    am_ = o["argmin"][b_, l]
                    print("image", b_, "light", l, "argmin>=0:", int((am_ >= 0).sum()))
            print("lights", lights.tolist(), "light_pt", o["light_pt"].cpu().numpy().tolist(), "f", f, "zoff", zoff)
    print(json.dumps({"cases": a.cases, "seed": a.seed, "worst_relative_difference": worst, "worst_cases": worst_case,
                      "gates": {"depth, albedo": 5e-6, "light, ambient": 5e-5},
                      "ok": all(v <= (5e-5 if ("light" in k or "ambient" in k) else 5e-6) for k, v in worst.items()), "seconds": time.time() - t0,
                      "library": L_.gcfr_version().decode()}))


if __name__ == "__main__":
    main()
