"""GPU parity at the configurations BASELINE.json names and at the edges of the exact-skip machinery.

  * minimum_distance + argmin of the reference's own torch.min (T8:514) from tests/golden/ -- the quantities every
    gradient is driven by, which the shadow weight w = tanh^2(d/2) only pins where d is small;
  * config 5 at full shape: one face, 18 lights, 512 x 512, 320 samples, against the C oracle;
  * config 3 at full batch: B = 32 fused forward + backward (render_from_depth), determinism within atomic
    jitter, gradients against autograd through the materialised oracle on a slice of the batch;
  * a 200-case slice of the randomised soak (tools/soak_parity.py) and adversarial surfaces for the depth-bound
    skip: plane slopes beyond the +-4 clamp, depth steps of 1e4, light distances below 60;
  * re-entrancy: two host threads, two streams, different gcfr_options, bit-equal to the serial results.
"""
import os
import sys
import threading

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from golden_cases import all_cases, H, W  # noqa: E402

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def camera(f, Hh, Ww):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = Ww / 2.0, Hh / 2.0
    return K


CASES = list(all_cases())


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_minimum_distance_and_argmin_match_the_reference(name, case):
    """HIP workspace kernel vs what the reference's torch.min returned.  Bit-equal except where torch-CPU's
    vectorised sqrt is off by an ulp from IEEE sqrtf (0.6 % of pixels, the same residue the C oracle shows); the
    index is identical wherever the distance bits agree."""
    from geomconsistentfr_amd import RenderParams, light_prep, shadow_min_distance
    prm, exp = case["params"], case["expect"]
    rp = RenderParams(n_samples=prm["n_samples"], t0=prm["t0"], dt=prm["dt"], light_distance=prm["light_distance"],
                      clamp_light_z_min=prm["clamp_light_z_min"])          # no bonus: the golden values precede it
    B = case["depth"].shape[0]
    _, pt = light_prep(to_dev(case["light"]), rp)
    md, am = shadow_min_distance(to_dev(case["depth"]), to_dev(case["mask"]), pt.reshape(B, 1, 3), rp)
    md, am = md[:, 0].cpu().numpy(), am[:, 0].cpu().numpy()
    md_ref, am_ref = exp["minimum_distance"], exp["argmin"].astype(np.int32)
    lit = md_ref < 1e5                                                     # "d < 1e6": the minimum is a real sample
    assert np.array_equal(lit, md < 1e5)
    np.testing.assert_array_equal(md[~lit], md_ref[~lit])                  # masked minimum: the reference's 1e6
    assert np.all(am[~lit] == -1)
    err = np.abs(md[lit] - md_ref[lit])
    assert err.max() <= 1e-5 * max(1.0, float(md_ref[lit].max())), err.max()
    assert (err == 0).mean() >= 0.98
    same_bits = lit & (md == md_ref)
    assert np.array_equal(am[same_bits], am_ref[same_bits])
    assert (am[lit] == am_ref[lit]).mean() >= 0.999


def test_config5_full_shape_18_lights_512_320():
    """BASELINE configs[4] on one GPU: one 512 x 512 face, 18 light directions, 320 march steps (1.51 G ray-steps)
    -- minimum distance and argmin bit-equal to the C oracle, fused shading within the north_star gates."""
    import scenes
    import c_oracle
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    S, L, N = 512, 18, 320
    depth, mask, albedo, normals, light, amb = scenes.synth_faces_sized(1, 3, S, L)
    prm = RenderParams(n_samples=N, dt=0.0025)
    out = R.render_fwd(to_dev(depth), to_dev(mask), to_dev(light), to_dev(amb), to_dev(normals), to_dev(albedo), prm,
                       want_argmin=True)
    _, pt_o = c_oracle.light_prep(light.reshape(-1, 3), clamp_z_min=0.0)
    tt = c_oracle.sample_table(0.025, 0.0025, N)
    np.testing.assert_array_equal(tt, np.arange(0.025, 0.825, 0.0025))
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o.reshape(1, L, 3), tt)
    md, am = out["minimum_distance"].cpu().numpy(), out["argmin"].cpu().numpy()
    assert np.array_equal(md, md_o)
    lit = md_o < 1e5
    assert np.array_equal(am[lit], am_o[lit]) and np.all(am[~lit] == -1)
    ref = c_oracle.shade(normals.astype(np.float64), depth, albedo, pt_o.reshape(1, L, 3), amb, md_o)
    assert np.abs(out["shadow_mask_weights"].cpu().numpy() - ref["shadow_w"]).max() <= 1e-6
    assert np.abs(out["rendered_images"].cpu().numpy() - ref["rendered"]).max() <= 1e-6
    # explicit default options give the same bits at this shape too (16 x 4 tiles, 4096 tiles per light)
    from geomconsistentfr_amd import _lib
    o2 = R.render_fwd(to_dev(depth), to_dev(mask), to_dev(light), to_dev(amb), to_dev(normals), to_dev(albedo), prm,
                      want_argmin=True, options=_lib.options())
    assert torch.equal(o2["minimum_distance"], out["minimum_distance"]) and torch.equal(o2["argmin"], out["argmin"])
    assert torch.equal(o2["rendered_images"], out["rendered_images"])


def test_config3_batch32_fused_forward_backward():
    """BASELINE configs[2]'s render block: B = 32 faces, 256 x 256 x 160, fused forward (normals + march + shading)
    and the one-launch fused backward.  Forward bit-deterministic, backward repeatable within f32 atomic jitter,
    batch-independent, and equal to autograd through the materialised oracle on two faces of the batch."""
    import scenes
    import materialised as M
    from normals_restatement import depth_to_normals
    from geomconsistentfr_amd.block import render_from_depth
    B = 32
    depth, mask, albedo, _normals, light, amb = scenes.synth_faces(B, 100)
    rng = np.random.default_rng(5)
    depth = depth + (2.0 * rng.random(depth.shape)).astype(np.float32)      # training-time depth is not smooth
    G_r = rng.random((B, 3, H, W), dtype=np.float32)
    G_w = rng.random((B, H, W), dtype=np.float32)
    K = camera(1570.0, H, W)

    def run(sel):
        leaves = [to_dev(a[sel]).requires_grad_() for a in (depth[:, None], albedo, light, amb)]
        o = render_from_depth(leaves[0], leaves[1], leaves[2], leaves[3], K.to(dev()), 1610.0, to_dev(mask[sel]))
        loss = (o["rendered_images"] * to_dev(G_r[sel])).sum() + (o["shadow_mask_weights"] * to_dev(G_w[sel])).sum()
        loss.backward()
        return o, [l.grad.cpu().numpy() for l in leaves]

    full = slice(0, B)
    o1, g1 = run(full)
    o2, g2 = run(full)
    for k in ("rendered_images", "shadow_mask_weights", "minimum_distance", "surface_normals"):
        assert torch.equal(o1[k], o2[k]), k
    for name, a, b in zip(("depth", "albedo", "light", "ambient"), g1, g2):
        assert np.abs(a - b).max() <= 2e-6 * max(np.abs(a).max(), 1e-6), name     # atomic-order jitter only
    # batch independence: faces 5..7 alone give the same gradients as inside the batch of 32
    _, g_sub = run(slice(5, 8))
    for name, a, b in zip(("depth", "albedo", "light", "ambient"), g1, g_sub):
        assert np.abs(a[5:8] - b).max() <= 2e-6 * max(np.abs(b).max(), 1e-6), name
    # against autograd through the materialised oracle (the reference's op graph), two faces of the batch
    for f in (0, 17):
        cl = [torch.from_numpy(a[f:f + 1]).clone().requires_grad_() for a in (depth[:, None], albedo, light, amb)]
        n = depth_to_normals(cl[0] + 1610.0, K)
        n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], 1)
        o = M.render_block(cl[0], cl[1], cl[2], cl[3], n, torch.from_numpy(mask[f:f + 1]))
        assert np.abs(o["rendered_images"].detach().numpy() - o1["rendered_images"][f:f + 1].detach().cpu().numpy()).max() <= 2e-5
        ((o["rendered_images"] * torch.from_numpy(G_r[f:f + 1])).sum()
         + (o["shadow_mask_weights"] * torch.from_numpy(G_w[f:f + 1])).sum()).backward()
        gd, ga, gl, gamb = [c.grad.numpy() for c in cl]
        scale = np.abs(gd).max()
        assert np.abs(g1[0][f:f + 1] - gd).max() <= 1e-3 * scale, f
        assert np.quantile(np.abs(g1[0][f:f + 1] - gd), 0.999) <= 1e-5 * scale, f
        assert np.abs(g1[1][f:f + 1] - ga).max() <= 1e-5, f
        np.testing.assert_allclose(g1[2][f:f + 1], gl, rtol=1e-4, atol=1e-4 * np.abs(gl).max())
        np.testing.assert_allclose(g1[3][f:f + 1], gamb, rtol=1e-4)


def test_soak_slice_is_bit_exact():
    """200 random (depth, mask, light) cases of tools/soak_parity.py (five sizes, depth scales 0.01 ... 300, light
    distances 30 ... 1e5): minimum distance AND argmin bit-equal to the C oracle on every unmasked pixel."""
    import soak_parity
    from geomconsistentfr_amd import _lib
    for opt in [None, _lib.options(tile_w=16, group=2)]:
        r = soak_parity.run_soak(200 if opt is None else 40, seed=20260928, options=opt)
        assert r["pixels_compared"] > (2_000_000 if opt is None else 300_000)
        assert r["lit_mask_mismatches"] == 0 and r["argmin_differences"] == 0, r
        assert r["max_abs_err_min_dist"] == 0.0, r
    # the inference kernels (no argmin)
    r = soak_parity.run_soak(120, seed=77, options=_lib.options(), want_argmin=False)
    assert r["pixels_compared"] > 1_000_000
    assert r["lit_mask_mismatches"] == 0 and r["max_abs_err_min_dist"] == 0.0, r


def _adversarial_surfaces(Hs, Ws, rng):
    r, c = np.mgrid[0:Hs, 0:Ws].astype(np.float64)
    x, y = c - Ws / 2.0, r - Hs / 2.0
    bump = 0.3 * Hs * np.exp(-((x / (0.25 * Ws)) ** 2 + (y / (0.3 * Hs)) ** 2))
    surf = {
        "slope6_x": 6.0 * x,                                                 # beyond the plane fit's +-4 slope clamp
        "slope10_xy": 10.0 * x - 7.5 * y + bump,
        "sawtooth": 40.0 * ((c % 16) / 16.0) + bump,                         # slope 2.5 with 40-high cliffs
        "step_1e4": np.where(c > Ws * 0.55, 1.0e4, 0.0) + bump,              # discontinuity of 1e4
        "step_-3e4_rows": np.where(r < Hs * 0.4, -3.0e4, 12.0) + 2 * rng.random((Hs, Ws)),
        "spikes": bump + 1.5e4 * (rng.random((Hs, Ws)) > 0.995),
        "huge_offset": bump + 2.0e5,
        "checker_1e3": 1.0e3 * (((r.astype(int) // 8) + (c.astype(int) // 8)) % 2) + bump,
    }
    return {k: v.astype(np.float32) for k, v in surf.items()}


@pytest.mark.parametrize("light_distance", [4013.0, 60.0, 30.0, 8.0])
def test_adversarial_surfaces_for_the_depth_bound_skip(light_distance):
    """Surfaces built to break the skip's hand-derived error bounds (slopes > 4, cliffs of 1e4+, spikes, offsets of
    2e5) under far and very near lights (distance 8 < the image half-width: rays far from parallel, light point
    inside the depth range): every tile shape and schedule must still give the C oracle's bits and argmin."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, light_prep, shadow_min_distance, _lib
    Hs, Ws, N = 128, 160, 96
    rng = np.random.default_rng(int(light_distance))
    surf = _adversarial_surfaces(Hs, Ws, rng)
    depth = np.stack(list(surf.values()))
    B = depth.shape[0]
    r, c = np.mgrid[0:Hs, 0:Ws]
    ell = ((((c - 0.5 * Ws) / (0.42 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.45 * Hs)) ** 2) < 1)
    mask = np.stack([ell if i % 2 == 0 else np.ones_like(ell) for i in range(B)]).astype(np.uint8)
    lights = np.array([[[0.75, 0.0, 0.66], [0.1, -0.2, 0.97], [-0.6, 0.7, 0.12]]] * B, np.float32)
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N, light_distance=light_distance)
    _, pt = light_prep(to_dev(lights), prm)
    pt_o = c_oracle.light_prep(lights.reshape(-1, 3), clamp_z_min=0.0, light_distance=light_distance)[1].reshape(B, 3, 3)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o, c_oracle.sample_table(0.025, 0.8 / N, N))
    lit = md_o < 1e5
    for tw in (0, 8, 16, 32, 64):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, options=_lib.options(tile_w=tw, ksplit=0))
        md, am = md.cpu().numpy(), am.cpu().numpy()
        bad = np.argwhere(md != md_o)
        assert bad.size == 0, (tw, list(surf)[bad[0][0]], bad[:3].tolist())
        assert np.array_equal(am[lit], am_o[lit]), tw


def test_two_host_threads_two_streams_different_options():
    """The library keeps no process-wide state (include/gcfr.h): two host threads launching on two streams with
    different gcfr_options, repeatedly and concurrently, give exactly the serial results."""
    from geomconsistentfr_amd import RenderParams, _lib
    from geomconsistentfr_amd import block as R
    rng = np.random.default_rng(77)
    B, L, Hs, Ws = 4, 2, 128, 128
    r, c = np.mgrid[0:Hs, 0:Ws]
    prm = RenderParams(n_samples=96, dt=0.008)
    opts = [_lib.options(tile_w=8, ksplit=0), _lib.options(tile_w=32, depth_bound_skip=0, group=2)]
    batches, refs = [], []
    for s in range(2):
        depth = (30 * np.exp(-(((c - 60 - 9 * s) / 30.0) ** 2 + ((r - 64) / 35.0) ** 2)) + rng.random((B, Hs, Ws))).astype(np.float32)
        mask = (rng.random((B, Hs, Ws)) > 0.2).astype(np.uint8)
        bt = [to_dev(depth), to_dev(mask), to_dev(rng.standard_normal((B, L, 3)).astype(np.float32)),
              to_dev((0.3 + 0.4 * rng.random((B, L))).astype(np.float32)),
              to_dev(rng.standard_normal((B, 3, Hs, Ws)).astype(np.float32)), to_dev(rng.random((B, 3, Hs, Ws)).astype(np.float32))]
        batches.append(bt)
        refs.append(R.render_fwd(*bt, prm, want_argmin=True))              # serial, default options
    torch.cuda.synchronize()
    plans = [R.RenderFwdPlan(B, L, Hs, Ws, prm, dev(), want_argmin=True, options=opts[s]) for s in range(2)]
    streams = [torch.cuda.Stream(device=dev()) for _ in range(2)]
    errors = []

    def worker(s):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(streams[s]):
                for _ in range(50):
                    plans[s](*batches[s])
            streams[s].synchronize()
        except Exception as e:                                              # noqa: BLE001
            errors.append(repr(e))

    threads = [threading.Thread(target=worker, args=(s,)) for s in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    torch.cuda.synchronize()
    assert not errors, errors
    for s in range(2):
        for k, v in refs[s].items():
            if v is not None:
                assert torch.equal(plans[s].out[k], v), (s, k)
