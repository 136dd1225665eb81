"""GPU: the opt-in "masked pixels only" render (gcfr_options.pixels = 1, RenderParams(pixels="mask")).

Pixels whose own mask cell is zero are not marched: they carry the value the reference gives a ray without an unmasked
sample (minimum distance 1e6, T8:512 -> shadow weight 1), every other pixel is bit-identical to the default.  Every consumer
in the training script multiplies by that mask (T8:619, 633, 641, 643), so the losses do not change by a bit."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, ROOT)
DEV = torch.device("cuda:0")


def _faces(B, rough):
    import scenes
    depth, mask, albedo, normals, light, amb = scenes.synth_faces(B, seed0=11)
    if rough:
        depth = depth + (400.0 * np.random.default_rng(3).random(depth.shape)).astype(np.float32)
    mask[1, 40:44, 200:230] = 1                      # an island far from the face: its pixels march, its neighbours do not
    mask[2] = 1                                      # an all-ones mask in the batch (no pixel is outside it)
    mask[3] = 0                                      # ... and an empty one (nothing marches)
    return depth, mask, albedo, normals, light, amb


@pytest.mark.parametrize("rough", [False, True])
def test_unmasked_pixels_are_bit_identical_and_masked_ones_carry_the_masked_value(rough):
    import c_oracle
    from geomconsistentfr_amd import RenderParams, light_prep, shadow_min_distance
    B = 6
    depth, mask, _, _, light, _ = _faces(B, rough)
    prm_all, prm_mask = RenderParams(), RenderParams(pixels="mask")
    _, pt = light_prep(torch.from_numpy(light).to(DEV), prm_all)
    d, m = torch.from_numpy(depth).to(DEV), torch.from_numpy(mask).to(DEV)
    md0, am0 = shadow_min_distance(d, m, pt.reshape(B, 1, 3), prm_all)
    md1, am1 = shadow_min_distance(d, m, pt.reshape(B, 1, 3), prm_mask)
    md0, am0, md1, am1 = (t.cpu().numpy()[:, 0] for t in (md0, am0, md1, am1))
    on = mask != 0
    assert np.array_equal(md1[on], md0[on]) and np.array_equal(am1[on], am0[on])
    assert (md1[~on] == 1e6).all() and (am1[~on] == -1).all()
    assert (md0[~on] < 1e5).any()                    # the default does march those pixels: the option is a real deviation there
    # ... and both against the C oracle
    _, pt_o = c_oracle.light_prep(light, clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(prm_all.t0, prm_all.dt, prm_all.n_samples))
    assert np.array_equal(md1[on], md_o[:, 0][on])
    lit = on & (md_o[:, 0] < 1e5)
    assert np.array_equal(am1[lit], am_o[:, 0][lit])


def test_fused_forward_with_the_option_shades_masked_pixels_unshadowed():
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    B = 4
    depth, mask, albedo, _, light, amb = _faces(B, False)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    cam = (1570.0, 1570.0, 128.0, 128.0, 1610.0)
    o0 = R.render_fwd(t(depth), t(mask), t(light).reshape(B, 1, 3), t(amb).reshape(B, 1), None, t(albedo), RenderParams(), want_argmin=True, camera=cam)
    o1 = R.render_fwd(t(depth), t(mask), t(light).reshape(B, 1, 3), t(amb).reshape(B, 1), None, t(albedo), RenderParams(pixels="mask"),
                      want_argmin=False, camera=cam)          # (the argmin plane is requested internally)
    on = torch.from_numpy(mask != 0).to(DEV)
    for k in ("minimum_distance", "shadow_mask_weights", "full_shading", "final_shading"):
        assert torch.equal(o1[k][:, 0][on], o0[k][:, 0][on]), k
    assert torch.equal(o1["rendered_images"][:, 0][on[:, None].expand(-1, 3, -1, -1)], o0["rendered_images"][:, 0][on[:, None].expand(-1, 3, -1, -1)])
    assert torch.equal(o1["full_shading"], o0["full_shading"]) and torch.equal(o1["surface_normals"], o0["surface_normals"])
    off = ~on
    assert (o1["shadow_mask_weights"][:, 0][off] == 1.0).all()
    assert torch.equal(o1["final_shading"][:, 0][off], o1["full_shading"][:, 0][off])
    assert (o1["argmin"][:, 0][off] == -1).all()


def test_training_losses_are_bit_equal_and_gradients_agree_with_the_option_on_and_off():
    """B = 32: generator and discriminator losses bit for bit; grad_albedo (plain stores) bit for bit; the atomically
    accumulated gradients (depth, and through it every network parameter) to within the run-to-run jitter of the float
    atomics' order, which is measured here by running the default twice."""
    from geomconsistentfr_amd.relightnet import PatchGAN, RelightNet
    from geomconsistentfr_amd.block import RenderParams
    from geomconsistentfr_amd.train import discriminator_losses, generator_losses, synthetic_batch
    B = 32
    torch.manual_seed(5)
    net = RelightNet().float().to(DEV)
    disc = PatchGAN().float().to(DEV)
    batch = synthetic_batch(B, 77, device=DEV)
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 1570.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    K = K.to(DEV)
    torch.backends.cudnn.deterministic = True            # (the convolutions' own run-to-run noise out of the comparison)
    torch.backends.cudnn.benchmark = False

    def run(pixels):
        net.render_params = RenderParams(pixels=pixels)
        net.zero_grad(set_to_none=True)
        with torch.no_grad():
            albedo, depth, SL = net.features(batch["images"], 200)
        leaves = [t.detach().clone().requires_grad_() for t in (albedo, depth, SL)]
        from geomconsistentfr_amd.block import render_from_depth
        r = render_from_depth(leaves[1], leaves[0], leaves[2][:, 0, 0, 1:4], leaves[2][:, 0, 0, 0], K, 1610.0,
                              batch["masks_fill"].reshape(B, 256, 256), net.render_params)
        out = (leaves[0], leaves[1], r["shadow_mask_weights"], r["ambient_light"], r["full_shading"], r["rendered_images"],
               r["unit_light_direction"], r["ambient_values"])
        img = batch["images"].permute(0, 3, 1, 2)
        m3 = batch["masks_fill"].permute(0, 3, 1, 2).repeat(1, 3, 1, 1)
        composite = out[5] * m3 + (1.0 - m3) * img
        with torch.no_grad():
            d_fake, d_real = discriminator_losses(disc, composite.detach(), img)
        L = generator_losses(out, batch, disc(composite))
        L["total"].backward()
        return ({k: v.detach().clone() for k, v in L.items()}, (d_fake.clone(), d_real.clone()),
                [t.grad.detach().clone() for t in leaves])

    L0, D0, G0 = run("all")
    L0b, D0b, G0b = run("all")
    L1, D1, G1 = run("mask")
    for k in L0:
        assert torch.equal(L0[k], L1[k]), (k, float(L0[k]), float(L1[k]))
    assert torch.equal(D0[0], D1[0]) and torch.equal(D0[1], D1[1])
    assert torch.equal(G0[0], G1[0])                      # albedo gradient: one plain store per pixel
    for g0, g0b, g1 in zip(G0[1:], G0b[1:], G1[1:]):     # depth (f32 atomics), light / ambient head (f64 atomics -> f32)
        jitter = float((g0 - g0b).abs().max())
        scale = float(g0.abs().max())
        assert float((g0 - g1).abs().max()) <= max(4.0 * jitter, 4e-6 * scale), (jitter, scale, float((g0 - g1).abs().max()))


def test_trainer_option_reaches_the_render_block():
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch
    torch.manual_seed(0)
    tr = Trainer(TrainConfig(miopen_find=False, render_pixels="mask"), device=DEV)
    assert tr.model.render_params.pixels == "mask"
    batch = synthetic_batch(4, 0, device=DEV)
    out = tr.model(batch["images"], 200, tr.K, batch["masks_fill"])
    off = batch["masks_fill"].reshape(4, 256, 256) == 0
    assert (out[2][off] == 1.0).all()                     # shadow_mask_weights outside the mask
    logs = tr.step(batch, 200, 0)
    assert all(np.isfinite(v) for v in logs.values())


def _golden_t8():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from golden_cases import t8_batches
    return list(t8_batches())


@pytest.mark.parametrize("name,case", _golden_t8(), ids=[n for n, _ in _golden_t8()])
def test_reference_golden_batches_through_the_option(name, case):
    """The reference's own T8 outputs (tests/golden/t8_*.npz) against the pixels = mask render: wherever the batch's mask is
    non-zero -- what every consumer of the training script keeps (T8:619, 633, 641, 643) -- shadow weight and RGB are within
    the same 2e-5 as the default path; outside the mask the option's documented values."""
    import dataclasses
    from geomconsistentfr_amd import RenderParams, render
    from normals_restatement import depth_to_normals
    prm, exp = case["params"], case["expect"]
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = prm["focal"]
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    n = depth_to_normals(torch.from_numpy(case["depth"])[:, None] + prm["normal_z_offset"], K)
    n[:, 1] = -n[:, 1]
    rp = dataclasses.replace(RenderParams(n_samples=prm["n_samples"], t0=prm["t0"], dt=prm["dt"], light_distance=prm["light_distance"],
                                          directional_intensity=prm["intensity"], clamp_light_z_min=prm["clamp_light_z_min"]),
                             pixels="mask")
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
    out = render(t(case["depth"])[:, None], t(case["albedo"]), t(case["light"]), t(case["ambient"]), n.float().to(DEV), t(case["mask"]), rp)
    on = case["mask"] != 0
    w = out["shadow_mask_weights"].cpu().numpy()
    if "shadow_mask_weights" in exp:
        assert np.abs(w - exp["shadow_mask_weights"])[on].max() <= 2e-5
    assert (w[~on] == 1.0).all()
    if "rendered_images" in exp:                         # (the shadow-path-only batches record no RGB)
        rgb = out["rendered_images"].cpu().numpy()
        on3 = np.repeat(on[:, None], 3, axis=1)
        assert np.abs(rgb - exp["rendered_images"])[on3].max() <= 2e-5
    if "rendered_images_face0" in exp:
        rgb = out["rendered_images"].cpu().numpy()[:1]
        assert np.abs(rgb - exp["rendered_images_face0"])[np.repeat(on[:1, None], 3, axis=1)].max() <= 2e-5


def test_non_finite_rays_are_nan_with_and_without_the_option():
    """A non-finite light point makes every ray of its image non-finite: minimum distance NaN (deviations), inside
    and outside the mask, whether or not pixels outside the mask are marched; the other images are unaffected."""
    from geomconsistentfr_amd import RenderParams, shadow_min_distance
    B = 4
    depth, mask, _, _, light, _ = _faces(B, False)
    pt = torch.from_numpy(4013.0 * light / np.linalg.norm(light, axis=1, keepdims=True)).float().to(DEV).reshape(B, 1, 3).contiguous()
    pt[1, 0, 0] = float("nan")
    d, m = torch.from_numpy(depth).to(DEV), torch.from_numpy(mask).to(DEV)
    md0, _ = shadow_min_distance(d, m, pt, RenderParams())
    md1, _ = shadow_min_distance(d, m, pt, RenderParams(pixels="mask"))
    md0, md1 = md0.cpu().numpy()[:, 0], md1.cpu().numpy()[:, 0]
    assert np.isnan(md0[1]).all() and np.isnan(md1[1]).all()
    on = mask != 0
    assert np.array_equal(md1[on], md0[on], equal_nan=True)
    assert (md1[0][~on[0]] == 1e6).all()
