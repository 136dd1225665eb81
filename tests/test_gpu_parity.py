"""GPU parity: the HIP path (through the C ABI, geomconsistentfr_amd._lib) against
  (1) the golden vectors generated from the reference (tests/golden/),
  (2) the CPU oracle on seeded inputs at other sizes / sample counts,
  (3) size-independent properties at BASELINE.json's full size (B=8, 256x256, N=160).
Gates (BASELINE.json north_star): shadow mask <= 1e-4 max-abs, shaded RGB <= 1e-3 max-abs.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from golden_cases import all_cases, H, W  # noqa: E402

pytestmark = pytest.mark.gpu

W_GATE = 1e-4      # north_star: shadow mask
RGB_GATE = 1e-3    # north_star: shaded RGB


def dev():
    return torch.device("cuda:0")


def to_dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def camera(f):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = W / 2.0
    K[:, 1, 2] = H / 2.0
    return K


def params_from(prm):
    from geomconsistentfr_amd import RenderParams
    return RenderParams(n_samples=prm["n_samples"], t0=prm["t0"], dt=prm["dt"],
                        light_distance=prm["light_distance"], directional_intensity=prm["intensity"],
                        clamp_light_z_min=prm["clamp_light_z_min"], inside_bonus=prm["bonus"],
                        bonus_box=prm["bonus_box"])


CASES = list(all_cases())


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_render_matches_reference_golden(name, case):
    from geomconsistentfr_amd import render
    from normals_restatement import depth_to_normals
    prm, exp = case["params"], case["expect"]
    n = depth_to_normals(torch.from_numpy(case["depth"])[:, None] + prm["normal_z_offset"], camera(prm["focal"]))
    n[:, 1] = -n[:, 1]
    out = render(to_dev(case["depth"])[:, None], to_dev(case["albedo"]), to_dev(case["light"]),
                 to_dev(case["ambient"]), n.float().to(dev()), to_dev(case["mask"]), params_from(prm))
    w = out["shadow_mask_weights"].cpu().numpy()
    if "shadow_mask_weights" in exp:                     # (t8_h pins the march's values / indices only: test_gpu_configs)
        e_w = np.abs(w - exp["shadow_mask_weights"]).max()
        assert e_w <= W_GATE, e_w
        assert e_w <= 2e-5, "inside the gate but worse than the expected fp32 noise: %g" % e_w
    np.testing.assert_allclose(out["unit_light_direction"].cpu().numpy().reshape(-1, 3),
                               exp["unit_light_direction"].reshape(-1, 3), atol=1e-7)
    if "full_shading" in exp:
        assert np.abs(out["full_shading"].cpu().numpy() - exp["full_shading"]).max() <= 1e-5
    if "rendered_images" in exp:
        e = np.abs(out["rendered_images"].cpu().numpy() - exp["rendered_images"]).max()
        assert e <= RGB_GATE, e
        assert e <= 2e-5, e
    if "rendered_images_face0" in exp:                   # the rough batch t8_f stores face 0's RGB only
        e = np.abs(out["rendered_images"].cpu().numpy()[:1] - exp["rendered_images_face0"]).max()
        assert e <= RGB_GATE, e
        assert e <= 2e-5, e


@pytest.mark.parametrize("Hs,Ws,N,t0,dt", [(64, 64, 48, 0.025, 0.016), (48, 96, 37, 0.02, 0.02),
                                            (130, 70, 160, 0.025, 0.005), (256, 256, 160, 0.025, 0.005),
                                            (512, 512, 320, 0.025, 0.0025), (1024, 2048, 12, 0.025, 0.06)])
def test_shadow_matches_c_oracle(Hs, Ws, N, t0, dt):
    """min distance + argmin against the C oracle, including non-tile-multiple sizes and config 5's size."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    rng = np.random.default_rng(Hs * 1000 + Ws)
    r, c = np.mgrid[0:Hs, 0:Ws]
    depth = (0.3 * Hs * np.exp(-(((c - 0.45 * Ws) / (0.2 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.25 * Hs)) ** 2))
             + rng.random((Hs, Ws))).astype(np.float32)
    mask = (rng.random((Hs, Ws)) > 0.15).astype(np.uint8)
    lights = np.array([[0.3, 0.5, 0.8], [-0.9, 0.1, 0.2], [0.001, -0.002, 1.0], [0.7, -0.7, 0.05],
                       [-0.6, 0.7, 0.1], [0.02, 0.9, 0.3], [0.5, -0.8, -0.3], [0.9, 0.05, 0.3]], np.float32)
    B = len(lights)
    prm = RenderParams(n_samples=N, t0=t0, dt=dt)
    unit_o, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
    unit, pt = light_prep(to_dev(lights), prm)
    np.testing.assert_array_equal(pt.cpu().numpy(), pt_o)
    depth_b = np.stack([np.roll(depth, 3 * b, axis=1) for b in range(B)])
    md, am = shadow_min_distance(to_dev(depth_b), to_dev(mask[None]), pt.reshape(B, 1, 3), prm)
    md_o, am_o = c_oracle.shadow_min_distance(depth_b, mask[None], pt_o[:, None, :], c_oracle.sample_table(t0, dt, N))
    md, am = md.cpu().numpy(), am.cpu().numpy()
    lit = md_o < 1e5
    assert np.array_equal(lit, md < 1e5)
    # bit for bit, as every sibling test asserts (the kernel follows the oracle's rounding sequence; a tolerance here
    # would hide the next regression): distances identical, argmin identical incl. ties (first minimal index)
    np.testing.assert_array_equal(md, md_o)
    # argmin: -1 marks "minimum is a masked sample" (no gradient)
    assert np.all(am[~lit] == -1)
    np.testing.assert_array_equal(am[lit], am_o[lit])


def _full_size_inputs(B=8):
    rng = np.random.default_rng(42)
    r, c = np.mgrid[0:H, 0:W]
    x, y = c - 128.0, r - 128.0
    depths, masks = [], []
    for b in range(B):
        ax, ay = 85 + 10 * rng.random(), 105 + 10 * rng.random()
        d = 80 * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0)) \
            + (30 + 10 * rng.random()) * np.exp(-(x ** 2 / 288 + (y - 12) ** 2 / 648)) + 3 * np.sin(c / 7) * np.cos(r / 9)
        depths.append(d.astype(np.float32))
        masks.append((((x / (ax - 8)) ** 2 + (y / (ay - 8)) ** 2) < 1).astype(np.uint8))
    lights = np.array([[0, .7071, .7071], [.8138, -.342, .4698], [-.8138, -.342, .4698], [.7518, 0, .6594],
                       [-.7076, .3892, .5897], [.4478, .4925, .7463], [-.5151, .4722, .7154], [.5145, 0, .8575]],
                      np.float32)[:B]
    return np.stack(depths), np.stack(masks), lights


def test_full_size_properties():
    """B=8 x 256x256 x 160 (BASELINE config 2): determinism, batch independence, mask semantics."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    prm = RenderParams()
    depth, mask, lights = _full_size_inputs()
    _, pt = light_prep(to_dev(lights), prm)
    md1, am1 = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(8, 1, 3), prm)
    md2, am2 = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(8, 1, 3), prm)
    assert torch.equal(md1, md2) and torch.equal(am1, am2)                       # deterministic
    for b in (0, 5):                                                             # batch independence
        mdb, _ = shadow_min_distance(to_dev(depth[b:b + 1]), to_dev(mask[b:b + 1]), pt[b].reshape(1, 1, 3), prm)
        assert torch.equal(mdb[0], md1[b])
    # multi-light layout (B=1, L=8) == batch layout with the same depth repeated
    md_l, _ = shadow_min_distance(to_dev(depth[:1]), to_dev(mask[:1]), pt.reshape(1, 8, 3), prm)
    md_b, _ = shadow_min_distance(to_dev(np.repeat(depth[:1], 8, 0)), to_dev(mask[:1]), pt.reshape(8, 1, 3), prm)
    assert torch.equal(md_l[0], md_b[:, 0])
    # all-zero mask: every sample is "outside the face" -> distance 1e6 everywhere (w = 1, fully lit)
    md0, am0 = shadow_min_distance(to_dev(depth), to_dev(np.zeros_like(mask)), pt.reshape(8, 1, 3), prm)
    assert torch.all(md0 == 1e6) and torch.all(am0 == -1)
    # enlarging the mask can only lower the minimum distance
    md_all, _ = shadow_min_distance(to_dev(depth), to_dev(np.ones_like(mask)), pt.reshape(8, 1, 3), prm)
    assert torch.all(md_all <= md1)
    assert torch.isfinite(md1).all() and (md1 >= 0).all()


def test_shade_matches_c_oracle_random_normals():
    import c_oracle
    from geomconsistentfr_amd.block import shade
    rng = np.random.default_rng(7)
    B, L, Hs, Ws = 2, 3, 40, 56
    normals = rng.standard_normal((B, 3, Hs, Ws)).astype(np.float32)
    depth = (30 * rng.random((B, Hs, Ws))).astype(np.float32)
    albedo = rng.random((B, 3, Hs, Ws), dtype=np.float32)
    pt = (4013 * rng.standard_normal((B, L, 3)) / 1.7).astype(np.float32)
    amb = rng.random((B, L), dtype=np.float32)
    md = (rng.random((B, L, Hs, Ws)) * 6).astype(np.float32)
    md[0, 0, :5] = 1e6
    md[1, 2, 5:9] = 0.0
    o = shade(to_dev(normals), to_dev(depth), to_dev(albedo), to_dev(pt), to_dev(amb), to_dev(md))
    ref = c_oracle.shade(normals.astype(np.float64), depth, albedo, pt, amb, md)
    assert np.abs(o["shadow_mask_weights"].cpu().numpy() - ref["shadow_w"]).max() <= 1e-6
    assert np.abs(o["full_shading"].cpu().numpy() - ref["full_shading"]).max() <= 1e-6
    assert np.abs(o["final_shading"].cpu().numpy() - ref["final_shading"]).max() <= 1e-6
    assert np.abs(o["rendered_images"].cpu().numpy() - ref["rendered"]).max() <= 1e-6


def test_no_cpu_path():
    from geomconsistentfr_amd import RenderParams, shadow_min_distance
    from geomconsistentfr_amd._lib import GcfrError
    with pytest.raises(GcfrError):
        shadow_min_distance(torch.zeros(1, 8, 8), torch.ones(1, 8, 8), torch.ones(1, 1, 3), RenderParams())


@pytest.mark.parametrize("Hs,Ws,N", [(64, 64, 48), (130, 70, 160), (48, 96, 37), (256, 256, 160), (258, 254, 33),
                                     (512, 512, 320)])
def test_workspace_kernel_is_bit_identical_to_direct_kernel(Hs, Ws, N):
    """The quad-texel / magic-rint / XCD-affine kernel must reproduce the direct-gather kernel bit for bit
    (values AND argmin), including odd half sizes (EVEN_HALF=false path) and B not a multiple of 8."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    rng = np.random.default_rng(Hs + 7 * Ws)
    B, L = 11, 2
    depth = (40 * rng.random((B, Hs, Ws))).astype(np.float32)
    mask = (rng.random((B, Hs, Ws)) > 0.3).astype(np.uint8)
    lights = rng.standard_normal((B, L, 3)).astype(np.float32)
    lights[0, 0] = (0.001, 0.002, 1.0)          # light projects inside the image
    lights[1, 1] = (0.0, 0.9, 0.1)              # purely vertical: column 0 exercises the -1 wrap
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
    _, pt = light_prep(to_dev(lights), prm)
    a_md, a_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=True)
    b_md, b_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=False)
    assert torch.equal(a_md, b_md)
    assert torch.equal(a_am, b_am)
    c_md, c_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, want_argmin=False, use_workspace=True)
    assert c_am is None and torch.equal(c_md, a_md)


@pytest.mark.parametrize("N", [160, 37, 5, 2])
def test_sample_range_split_is_bit_identical(N):
    """The k-split variant (4 waves share a tile, a quarter of the samples each, LDS combine) against the
    one-tile-per-wave variant and the direct kernel: same bits, same argmin (ties -> earliest index)."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep, _lib
    rng = np.random.default_rng(N)
    B, Hs, Ws = 3, 66, 130
    depth = (40 * rng.random((B, Hs, Ws))).astype(np.float32)
    depth[1] = 7.0                                      # flat depth -> many exact ties between samples
    mask = (rng.random((B, Hs, Ws)) > 0.3).astype(np.uint8)
    lights = rng.standard_normal((B, 2, 3)).astype(np.float32)
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
    _, pt = light_prep(to_dev(lights), prm)
    ref_md, ref_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=False)
    for ks in (0, 1):                                   # one tile per wave (the grid), k-split
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=True, options=_lib.options(ksplit=ks))
        assert torch.equal(md, ref_md), ks
        assert torch.equal(am, ref_am), ks


def _compact_masks(Hs, Ws):
    r, c = np.mgrid[0:Hs, 0:Ws]
    masks = {
        "ellipse": ((((c - 0.55 * Ws) / (0.3 * Ws)) ** 2 + ((r - 0.45 * Hs) / (0.35 * Hs)) ** 2) < 1),
        "small_rect": (r >= 10) & (r < 14) & (c >= Ws - 9) & (c < Ws - 3),
        "single_pixel": (r == Hs // 3) & (c == Ws // 4),
        "empty": np.zeros((Hs, Ws), bool),
        "border_rows": (r < 2) | (r >= Hs - 1),
        "left_column": (c == 0),
        "two_blobs": ((((c - 12) ** 2 + (r - 12) ** 2) < 30) | (((c - Ws + 10) ** 2 + (r - Hs + 14) ** 2) < 40)),
    }
    return {k: v.astype(np.uint8) for k, v in masks.items()}


@pytest.mark.parametrize("Hs,Ws,N", [(96, 128, 160), (130, 70, 37), (256, 256, 160)])
def test_mask_bounding_box_pruning_is_exact(Hs, Ws, N):
    """Compact / degenerate masks: the workspace kernel bounds each wave's sample loop by the mask's bounding
    box; values and argmin must stay bit-identical to the direct kernel, with and without k-split."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep, _lib
    rng = np.random.default_rng(Hs + Ws + N)
    masks = _compact_masks(Hs, Ws)
    B = len(masks)
    mask = np.stack(list(masks.values()))
    r, c = np.mgrid[0:Hs, 0:Ws]
    depth = np.stack([(0.3 * Hs * np.exp(-(((c - 0.5 * Ws) / (0.25 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.3 * Hs)) ** 2))
                       + rng.random((Hs, Ws))).astype(np.float32) for _ in range(B)])
    lights = np.array([[0.3, 0.5, 0.8], [-0.9, 0.1, 0.2], [0.001, -0.002, 1.0], [0.7, -0.7, 0.05], [0.0, 0.9, 0.1],
                       [0.9, 0.0, 0.1]], np.float32)
    lights = np.stack([np.roll(lights, b, axis=0)[:3] for b in range(B)])                  # (B,3,3)
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.8 / N)
    _, pt = light_prep(to_dev(lights), prm)
    ref_md, ref_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=False)
    for ks in (0, 1):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=True, options=_lib.options(ksplit=ks))
        bad = (md != ref_md).nonzero()
        assert torch.equal(md, ref_md), (ks, bad[:5].tolist())
        assert torch.equal(am, ref_am), ks
    empty = list(masks).index("empty")
    assert torch.all(ref_md[empty] == 1e6)
    # one mask shared by the whole batch (S1 form): bounding box of mask 0 applies to every image
    md1, am1 = shadow_min_distance(to_dev(depth), to_dev(mask[:1]), pt, prm, use_workspace=True)
    md1r, am1r = shadow_min_distance(to_dev(depth), to_dev(mask[:1]), pt, prm, use_workspace=False)
    assert torch.equal(md1, md1r) and torch.equal(am1, am1r)


def test_non_finite_inputs_do_not_fault_and_stay_local():
    """NaN / inf depth and degenerate lights: no GPU fault (raw buffer range checks), finite pixels elsewhere
    unaffected where the reference's data flow says so, NaN light -> NaN outputs (visible, not silent)."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    prm = RenderParams()
    rng = np.random.default_rng(3)
    Hs = Ws = 128
    depth = (20 * rng.random((3, Hs, Ws))).astype(np.float32)
    mask = np.ones((3, Hs, Ws), np.uint8)
    clean = depth.copy()
    depth[0, 40, 50] = np.nan
    depth[1, 10:12, 100] = np.inf
    lights = np.array([[0.3, 0.5, 0.8], [0.0, 0.0, 0.0], [np.nan, 0.2, 0.5]], np.float32)
    unit, pt = light_prep(to_dev(lights), prm)
    assert torch.all(unit[1] == 0) and torch.all(pt[1] == 0)             # normalize(0) = 0 (SLT pass 1, SLT:543)
    md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(3, 1, 3), prm)
    torch.cuda.synchronize()
    md_clean, _ = shadow_min_distance(to_dev(clean), to_dev(mask), pt.reshape(3, 1, 3), prm)
    md, md_clean = md.cpu().numpy(), md_clean.cpu().numpy()
    # image 0: only rays that sample the NaN texel (or start on it) can differ; far-away pixels are untouched
    same = (md[0, 0] == md_clean[0, 0])
    assert same.mean() > 0.9 and np.isnan(md[0, 0]).sum() <= (~same).sum()
    assert np.isfinite(md[1, 0]).mean() > 0.9
    assert np.isnan(md[2, 0]).all()                                       # NaN light poisons its own image only
    assert np.isfinite(md_clean[:2]).all()


def test_python_boundary_rejects_bad_shapes():
    from geomconsistentfr_amd import RenderParams, render, shadow_min_distance
    from geomconsistentfr_amd._lib import GcfrError
    d = torch.zeros(2, 1, 64, 64, device="cuda:0")
    with pytest.raises((GcfrError, RuntimeError, ValueError)):
        render(d, torch.zeros(2, 3, 64, 64, device="cuda:0"), torch.zeros(3, 3, device="cuda:0"),
               torch.zeros(2, device="cuda:0"), torch.zeros(2, 3, 64, 64, device="cuda:0"),
               torch.ones(2, 64, 64, device="cuda:0"))
    with pytest.raises(GcfrError):                                        # odd width is outside the supported set
        shadow_min_distance(torch.zeros(1, 64, 63, device="cuda:0"), torch.ones(1, 64, 63, device="cuda:0"),
                            torch.ones(1, 1, 3, device="cuda:0"), RenderParams())
    with pytest.raises(GcfrError):                                        # mask batch must be 1 or B
        shadow_min_distance(torch.zeros(3, 64, 64, device="cuda:0"), torch.ones(2, 64, 64, device="cuda:0"),
                            torch.ones(3, 1, 3, device="cuda:0"), RenderParams())


def test_fractional_float_masks_mean_non_zero():
    """The reference's masks are imread / 255.0 floats with four grey levels (64/255, 128/255, ...) and it tests
    `mask == 0` (T8:510): every non-zero level is inside the face.  Every entry point converts with `!= 0`, never
    by truncation -- eager call, plan call and a hipGraph replay after new data was copied into the captured inputs."""
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    rng = np.random.default_rng(9)
    B, Hs, Ws = 2, 64, 64
    mk = lambda *s: to_dev(rng.random(s, dtype=np.float32))
    depth, albedo, normals = to_dev((30 * rng.random((B, Hs, Ws))).astype(np.float32)), mk(B, 3, Hs, Ws), mk(B, 3, Hs, Ws) - 0.5
    levels = np.array([0.0, 64 / 255.0, 128 / 255.0, 192 / 255.0, 1.0])
    mask_f = levels[rng.integers(0, 5, (B, Hs, Ws))]
    light, amb = to_dev(rng.standard_normal((B, 1, 3)).astype(np.float32)), mk(B, 1)
    prm = RenderParams(n_samples=40, dt=0.02)
    ref = R.render_fwd(depth, to_dev((mask_f != 0).astype(np.uint8)), light, amb, normals, albedo, prm, want_argmin=False)
    for mdt in (np.float64, np.float32):
        out = R.render_fwd(depth, to_dev(mask_f.astype(mdt)), light, amb, normals, albedo, prm, want_argmin=False)
        assert torch.equal(out["rendered_images"], ref["rendered_images"])
        assert torch.equal(out["minimum_distance"], ref["minimum_distance"])
    assert R.mask_to_u8(to_dev(mask_f)).dtype == torch.uint8
    assert int(R.mask_to_u8(to_dev(mask_f)).sum()) == int((mask_f != 0).sum())
    # graph replay: new inputs are copied into the captured tensors (the mask through mask_to_u8), then replayed
    plan = R.RenderFwdPlan(B, 1, Hs, Ws, prm, depth.device, want_argmin=False)
    static = [depth.clone(), torch.zeros((B, Hs, Ws), dtype=torch.uint8, device=depth.device), light.clone(),
              amb.clone(), normals.clone(), albedo.clone()]
    plan.capture(*static)
    static[1].copy_(R.mask_to_u8(to_dev(mask_f)))
    out = plan.replay()
    torch.cuda.synchronize()
    assert torch.equal(out["rendered_images"], ref["rendered_images"])


def test_maximum_supported_size():
    """4096 x 4096 (the ABI's upper bound) with a short sample table: bit-equal to the oracle, both kernels."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    Hs = Ws = 4096
    rng = np.random.default_rng(4096)
    depth = (50 * rng.random((1, Hs, Ws), dtype=np.float32))
    r, c = np.ogrid[0:Hs, 0:Ws]
    mask = ((((c - 2000) / 1500.0) ** 2 + ((r - 2100) / 1800.0) ** 2) < 1).astype(np.uint8)[None]
    light = np.array([[0.4, -0.3, 0.6]], np.float32)
    prm = RenderParams(n_samples=4, t0=0.025, dt=0.2)
    _, pt = light_prep(to_dev(light), prm)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, c_oracle.light_prep(light, clamp_z_min=0.0)[1][:, None, :],
                                              c_oracle.sample_table(0.025, 0.2, 4))
    for ws in (True, False):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(1, 1, 3), prm, use_workspace=ws)
        assert np.array_equal(md.cpu().numpy(), md_o), ws
        lit = md_o < 1e5
        assert np.array_equal(am.cpu().numpy()[lit], am_o[lit]), ws


@pytest.mark.parametrize("Hs,Ws,N,dt", [(128, 128, 160, 0.005), (66, 130, 37, 0.02), (256, 256, 96, 0.008)])
def test_depth_bound_skip_is_exact_for_every_tile_shape(Hs, Ws, N, dt):
    """The depth-bound group skip (plane-band depth-bounds grid, gcfr_options.depth_bound_skip) must not change one bit:
    min distance and argmin against the C oracle and the direct kernel, for every tile shape and group size,
    on surfaces that make it fire (smooth bump), that defeat it (noise), that straddle zero and that are
    offset far from zero (the sampled-zero quirk of integral coordinates, coarse float spacing -> ties)."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep, _lib
    rng = np.random.default_rng(Hs * 7 + N)
    r, c = np.mgrid[0:Hs, 0:Ws]
    bump = 0.35 * Hs * np.exp(-(((c - 0.5 * Ws) / (0.25 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.3 * Hs)) ** 2))
    depth = np.stack([bump, bump + 3 * rng.random((Hs, Ws)), 30 * rng.random((Hs, Ws)), bump - 0.15 * Hs,
                      -bump, bump + 1000.0, np.round(bump)]).astype(np.float32)
    B = depth.shape[0]
    ell = ((((c - 0.5 * Ws) / (0.4 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.45 * Hs)) ** 2) < 1)
    mask = np.stack([ell, ell, np.ones_like(ell), ell, ell, ell, rng.random((Hs, Ws)) > 0.2]).astype(np.uint8)
    lights = np.array([[[0.75, 0.0, 0.66], [0.1, -0.2, 0.97]]] * B, np.float32)
    lights[1::2, 0] = [-0.5, 0.47, 0.72]
    lights[2, 1] = [0.99, 0.05, 0.05]                                  # grazing
    prm = RenderParams(n_samples=N, t0=0.025, dt=dt)
    _, pt = light_prep(to_dev(lights), prm)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, c_oracle.light_prep(lights.reshape(-1, 3), clamp_z_min=0.0)[1]
                                              .reshape(B, 2, 3), c_oracle.sample_table(0.025, dt, N))
    lit = md_o < 1e5
    ref_md, ref_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=False)
    assert np.array_equal(ref_md.cpu().numpy(), md_o)
    assert np.array_equal(ref_am.cpu().numpy()[lit], am_o[lit])
    for zb in (1, 0):
        for tw in (8, 16, 32, 64):
            for grp in ((4, 2, 1) if tw == 8 else (4,)):
                opt = _lib.options(depth_bound_skip=zb, tile_w=tw, group=grp, ksplit=0)
                md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=True, options=opt)
                assert torch.equal(md, ref_md), (zb, tw, grp)
                assert torch.equal(am, ref_am), (zb, tw, grp)


def test_render_fwd_plan_matches_eager_call_and_overlaps_on_two_streams():
    """RenderFwdPlan (preallocated outputs, one ctypes call) gives the bits of render_fwd; two plans driven
    round-robin on two streams (bench.py's default) do not disturb each other."""
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(5)
    B, L, Hs, Ws = 3, 2, 96, 128
    r, c = np.mgrid[0:Hs, 0:Ws]
    batches = []
    for s in range(2):
        depth = (30 * np.exp(-(((c - 60 - 5 * s) / 30.0) ** 2 + ((r - 50) / 35.0) ** 2)) + rng.random((B, Hs, Ws))).astype(np.float32)
        mask = (rng.random((B, Hs, Ws)) > 0.2).astype(np.uint8)
        light = rng.standard_normal((B, L, 3)).astype(np.float32)
        amb = (0.3 + 0.4 * rng.random((B, L))).astype(np.float32)
        nrm = rng.standard_normal((B, 3, Hs, Ws)).astype(np.float32)
        alb = rng.random((B, 3, Hs, Ws)).astype(np.float32)
        batches.append([torch.from_numpy(t).to(dev) for t in (depth, mask, light, amb, nrm, alb)])
    prm = RenderParams(n_samples=64, dt=0.0125)
    ref = [R.render_fwd(*bt, prm, want_argmin=True) for bt in batches]
    plans = [R.RenderFwdPlan(B, L, Hs, Ws, prm, dev, want_argmin=True) for _ in range(2)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    torch.cuda.synchronize()
    for it in range(6):
        with torch.cuda.stream(streams[it % 2]):
            plans[it % 2](*batches[it % 2])
    torch.cuda.synchronize()
    for s in range(2):
        for k, v in ref[s].items():
            if v is not None:
                assert torch.equal(plans[s].out[k], v), (s, k)
    # hipGraph capture of a plan call: replays give the same bits, also interleaved on two streams, and pick up
    # new data written into the captured input tensors
    gplans = [R.RenderFwdPlan(B, L, Hs, Ws, prm, dev, want_argmin=True).capture(*[t.clone() for t in batches[s]])
              for s in range(2)]
    for it in range(6):
        with torch.cuda.stream(streams[it % 2]):
            gplans[it % 2].replay()
    torch.cuda.synchronize()
    for s in range(2):
        for k, v in ref[s].items():
            if v is not None:
                assert torch.equal(gplans[s].out[k], v), (s, k)
    for dst, src in zip(gplans[0]._static, batches[1]):
        dst.copy_(src)
    gplans[0].replay()
    torch.cuda.synchronize()
    for k, v in ref[1].items():
        if v is not None:
            assert torch.equal(gplans[0].out[k], v), k


@pytest.mark.parametrize("t0,dt,N", [(0.82, -0.005, 160), (0.4, 0.0, 8), (0.025, 0.005, 1)])
def test_non_increasing_sample_tables_fall_back_to_the_full_march(t0, dt, N):
    """The exact pruning / skip tests assume an increasing table; a decreasing or constant one (or N = 1) must
    still give the direct kernel's and the oracle's bits."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    rng = np.random.default_rng(N)
    Hs, Ws = 96, 128
    r, c = np.mgrid[0:Hs, 0:Ws]
    depth = (30 * np.exp(-(((c - 60) / 30.0) ** 2 + ((r - 50) / 35.0) ** 2)) + rng.random((2, Hs, Ws))).astype(np.float32)
    mask = np.stack([(((c - 64) / 50.0) ** 2 + ((r - 48) / 40.0) ** 2) < 1, rng.random((Hs, Ws)) > 0.3]).astype(np.uint8)
    lights = np.array([[0.6, 0.2, 0.7], [-0.4, -0.5, 0.6]], np.float32)
    prm = RenderParams(n_samples=N, t0=t0, dt=dt)
    _, pt = light_prep(to_dev(lights), prm)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, c_oracle.light_prep(lights, clamp_z_min=0.0)[1][:, None, :],
                                              c_oracle.sample_table(t0, dt, N))
    lit = md_o < 1e5
    for ws in (False, True):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(2, 1, 3), prm, use_workspace=ws)
        assert np.array_equal(md.cpu().numpy(), md_o), ws
        assert np.array_equal(am.cpu().numpy()[lit], am_o[lit]), ws


@pytest.mark.parametrize("Hs,Ws,N,dt", [(64, 96, 40, 0.02), (128, 128, 80, 0.01), (256, 256, 160, 0.005), (96, 64, 33, 0.024)])
def test_lds_staged_march_is_bit_identical(Hs, Ws, N, dt):
    """gcfr_options.lds_stage = 1 (round 3): the workgroup's mask -- as a bitmap -- and its depth-bounds records are read
    from LDS instead of being gathered through the texture path.  Same cells, same records, same arithmetic: min_dist
    and argmin must equal the global-path kernel's and the C oracle's bits -- smooth / rough / offset / negative surfaces,
    ellipse, all-ones and random masks, one mask shared by the batch, with and without the depth-bound skip, the
    inference and the training (argmin) kernels, stand-alone and with the fused shading epilogue."""
    import c_oracle
    from geomconsistentfr_amd import RenderParams, _lib, light_prep, shadow_min_distance
    from geomconsistentfr_amd.block import render_fwd
    rng = np.random.default_rng(Hs * 7 + Ws)
    r, c = np.mgrid[0:Hs, 0:Ws]
    bump = 0.35 * Hs * np.exp(-(((c - 0.5 * Ws) / (0.25 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.3 * Hs)) ** 2))
    depth = np.stack([bump, bump + 3 * rng.random((Hs, Ws)), 30 * rng.random((Hs, Ws)), bump - 0.15 * Hs, -bump,
                      bump + 1000.0]).astype(np.float32)
    B = depth.shape[0]
    ell = ((((c - 0.5 * Ws) / (0.4 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.45 * Hs)) ** 2) < 1)
    mask = np.stack([ell, ell, np.ones_like(ell), rng.random((Hs, Ws)) > 0.2, ell, rng.random((Hs, Ws)) > 0.7]).astype(np.uint8)
    lights = np.array([[[0.75, 0.0, 0.66], [0.1, -0.2, 0.97]]] * B, np.float32)
    lights[1::2, 0] = [-0.5, 0.47, 0.72]
    lights[2, 1] = [0.99, 0.05, 0.05]                                  # grazing
    prm = RenderParams(n_samples=N, t0=0.025, dt=dt)
    _, pt = light_prep(to_dev(lights), prm)
    tt = c_oracle.sample_table(0.025, dt, N)
    pt_o = c_oracle.light_prep(lights.reshape(-1, 3), clamp_z_min=0.0)[1].reshape(B, 2, 3)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o, tt)
    lit = md_o < 1e5
    for zb in (1, 0):
        for want in (True, False):
            md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, want_argmin=want, use_workspace=True,
                                         options=_lib.options(lds_stage=1, ksplit=0, depth_bound_skip=zb))
            assert np.array_equal(md.cpu().numpy(), md_o), (zb, want)
            if want:
                assert np.array_equal(am.cpu().numpy()[lit], am_o[lit]), zb
    # one mask for the whole batch (mask_batch = 1): per-image bitmaps collapse to one
    md_s, _ = c_oracle.shadow_min_distance(depth, np.repeat(mask[3:4], B, 0), pt_o, tt)
    md, _ = shadow_min_distance(to_dev(depth), to_dev(mask[3:4]), pt, prm, want_argmin=False, use_workspace=True,
                                options=_lib.options(lds_stage=1, ksplit=0))
    assert np.array_equal(md.cpu().numpy(), md_s)
    # fused epilogue: every output of the two variants is the same bits
    amb = to_dev((0.3 + 0.4 * rng.random((B, 2))).astype(np.float32))
    nrm = to_dev(rng.standard_normal((B, 3, Hs, Ws)).astype(np.float32))
    alb = to_dev(rng.random((B, 3, Hs, Ws)).astype(np.float32))
    outs = [render_fwd(to_dev(depth), to_dev(mask), to_dev(lights), amb, nrm, alb, prm, want_argmin=True,
                       options=_lib.options(lds_stage=ls, ksplit=0)) for ls in (0, 1)]
    for k in outs[0]:
        if outs[0][k] is not None:
            assert torch.equal(outs[0][k], outs[1][k]), k


@pytest.mark.parametrize("lds", [0, 1])
def test_non_finite_inputs_never_make_the_skipping_kernels_differ_from_the_plain_one(lds):
    """Round-2 advisor note: with a checked sample table the depth-bounds record offset is used unchecked (`zb_trusted`);
    a raw buffer load answers an out-of-range offset with zeros -- the record "surface is exactly z = 0", a wrong skip --
    so exactness rests on every rounded cell lying inside the image.  It does whatever the inputs: samples sit on the
    segment pixel -> clamped end point, and a non-finite end point (NaN / inf light, `finite_ray` false) collapses the
    segment onto the pixel itself.  Adversarial inputs -- NaN / +-inf depth cells, NaN and inf light components, a
    light exactly above the image centre -- through the workspace kernels (bounds on and off, global and LDS-staged)
    against the plain direct-gather kernel, which has no bounds machinery: identical bits, NaNs in the same places."""
    from geomconsistentfr_amd import RenderParams, _lib, shadow_min_distance
    rng = np.random.default_rng(77)
    Hs, Ws, N = 128, 128, 80
    r, c = np.mgrid[0:Hs, 0:Ws]
    base = (40 * np.exp(-(((c - 64) / 30.0) ** 2 + ((r - 64) / 35.0) ** 2)) + rng.random((Hs, Ws))).astype(np.float32)
    depth = np.stack([base] * 6)
    depth[1, 30:34, 40:44] = np.nan
    depth[2, 90, 17] = np.inf
    depth[2, 20, 100] = -np.inf
    depth[3, ::16, ::16] = np.nan
    depth[4, 64, 64] = 1e30
    mask = (rng.random((6, Hs, Ws)) > 0.15).astype(np.uint8)
    pt = np.array([[[3000.0, 500.0, 2600.0]], [[-2500.0, 1500.0, 2700.0]], [[100.0, -3500.0, 1900.0]],
                   [[np.nan, 100.0, 3000.0]], [[2000.0, np.inf, 3000.0]], [[0.0, 0.0, 4013.0]]], np.float32)
    prm = RenderParams(n_samples=N, t0=0.025, dt=0.01)
    ref_md, ref_am = shadow_min_distance(to_dev(depth), to_dev(mask), to_dev(pt), prm, use_workspace=False)
    for zb in (1, 0):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), to_dev(pt), prm, use_workspace=True,
                                     options=_lib.options(ksplit=0, depth_bound_skip=zb, lds_stage=lds))
        a, b = md.cpu().numpy(), ref_md.cpu().numpy()
        np.testing.assert_array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(b)
        assert np.array_equal(a[ok], b[ok]), (zb, lds)
        assert np.array_equal(am.cpu().numpy()[ok], ref_am.cpu().numpy()[ok]), (zb, lds)
    assert np.isnan(ref_md.cpu().numpy()[3]).all() and np.isfinite(ref_md.cpu().numpy()[0]).all()


@pytest.mark.parametrize("knobs", [dict(ksplit=0), dict(ksplit=0, depth_bound_skip=0), dict(ksplit=0, tile_w=8, group=2),
                                   dict(ksplit=0, tile_w=32, depth_bound_skip=0)])
def test_rough_depth_marches_bit_identically_through_the_rough_loop(knobs):
    """Round 4: tiles that march without the depth bounds -- they gave them up (depth rougher than the rays rise: noise of
    amplitude 400, an untrained network's output), or the caller switched the bounds off -- run the rough loop (one position per
    sample, mask byte and texel gathered together, no group bookkeeping).  Against the C oracle, both marches, bit for bit;
    image 2 is smooth on its left half and rough on its right (both variants in one launch), image 3 has an all-ones mask."""
    import c_oracle
    from geomconsistentfr_amd import _lib, RenderParams, shadow_min_distance, light_prep
    depth, mask, lights = _full_size_inputs(4)
    rng = np.random.default_rng(9)
    noise = (400.0 * rng.random(depth.shape)).astype(np.float32)
    noise[2, :, :128] = 0.0
    depth = depth + noise
    mask[3] = 1
    prm = RenderParams()
    _, pt = light_prep(to_dev(lights), prm)
    _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(prm.t0, prm.dt, prm.n_samples))
    lit = md_o < 1e5
    for want_argmin in (False, True):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt.reshape(4, 1, 3), prm, want_argmin=want_argmin,
                                     options=_lib.options(**knobs))
        np.testing.assert_array_equal(md.cpu().numpy(), md_o)
        if want_argmin:
            np.testing.assert_array_equal(am.cpu().numpy()[lit], am_o[lit])


@pytest.mark.parametrize("Hs,Ws", [(362, 362), (362, 364), (368, 400), (500, 500), (512, 512), (514, 512)])
def test_statistics_chunks_around_the_doubled_size_class_are_bit_identical(Hs, Ws):
    """The prepass' per-image statistics (mask box / octagon, all-ones flag, depth range) are partial records per chunk of
    16,384 pixels -- 32,768 for images of 9 ... 16 plain chunks, so that a 512 x 512 image has eight records the march folds on
    the scalar unit (csrc/gcfr_march.hpp stat_chunk_px).  Sizes on both sides of both class borders, widths with and without
    the 16-pixel vector path, and masks / depths whose deciding cells lie in the SECOND half of a doubled chunk or in the last,
    partial one: a record that misses them gives a wrong box (samples skipped), a wrong all-ones flag or a depth range that
    is too narrow (bounds that are none) -- the direct kernel uses none of them."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance, light_prep
    rng = np.random.default_rng(Hs * 7 + Ws)
    P = Hs * Ws
    r, c = np.mgrid[0:Hs, 0:Ws]
    base = (0.25 * Hs * np.exp(-(((c - 0.5 * Ws) / (0.3 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.3 * Hs)) ** 2))).astype(np.float32)
    masks, depths = [], []
    def flat_cells(lo, hi):
        m = np.zeros(P, np.uint8)
        m[lo:hi] = 1
        return m.reshape(Hs, Ws)
    masks.append(flat_cells(16384 + 40, 32768 - 40))                 # only cells of the second 16,384 pixels
    masks.append(flat_cells(P - 3 * Ws - 7, P))                      # only cells of the last (partial) chunk
    m = np.ones(P, np.uint8); m[16384 + 5] = 0; masks.append(m.reshape(Hs, Ws))      # all ones but one cell, second half
    m = np.ones(P, np.uint8); m[P - 1] = 0; masks.append(m.reshape(Hs, Ws))          # ... but the very last cell
    masks.append(np.ones((Hs, Ws), np.uint8))                        # all ones
    masks.append((((c - 0.55 * Ws) / (0.3 * Ws)) ** 2 + ((r - 0.6 * Hs) / (0.35 * Hs)) ** 2 < 1).astype(np.uint8))
    B = len(masks)
    for b in range(B):
        d = (base + rng.random((Hs, Ws)).astype(np.float32)).reshape(P)
        if b % 2 == 0:
            d[16384 + 9 * (b + 1)] = 4000.0 + b                      # the image's depth maximum / minimum live in the second half
            d[32768 - 11 * (b + 1)] = -3000.0 - b
        else:
            d[P - 2 - b] = 2500.0                                    # ... or in the last chunk
        depths.append(d.reshape(Hs, Ws))
    mask, depth = np.stack(masks), np.stack(depths)
    lights = np.array([[0.3, 0.5, 0.8], [-0.9, 0.1, 0.2], [0.7, -0.7, 0.05]], np.float32)
    lights = np.stack([np.roll(lights, b, axis=0) for b in range(B)])
    prm = RenderParams(n_samples=40, t0=0.025, dt=0.02)
    _, pt = light_prep(to_dev(lights), prm)
    ref_md, ref_am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, use_workspace=False)
    for want in (True, False):
        md, am = shadow_min_distance(to_dev(depth), to_dev(mask), pt, prm, want_argmin=want, use_workspace=True)
        assert torch.equal(md, ref_md), (want, (md != ref_md).nonzero()[:5].tolist())
        if want:
            assert torch.equal(am, ref_am)
    import c_oracle
    md_o, _ = c_oracle.shadow_min_distance(depth[:2], mask[:2], pt[:2].cpu().numpy(), c_oracle.sample_table(0.025, 0.02, 40))
    np.testing.assert_array_equal(ref_md[:2].cpu().numpy(), md_o)
