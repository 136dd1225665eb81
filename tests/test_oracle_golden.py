"""CPU: the C oracle and the materialised torch port reproduce the reference's golden vectors.

This is what pins the oracle (task brief (3)): every fixture under tests/golden/ was produced by the
unmodified reference (oracle/make_golden.py).  Tolerances: shadow weight <= 2e-6 (the reference's CPU
run uses torch's vectorised sqrt/exp, which are not correctly rounded; the oracle uses libm), shading
<= 1e-6 -- far inside north_star's 1e-4 / 1e-3 gates.
"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import c_oracle  # noqa: E402
import materialised as M  # noqa: E402
from normals_restatement import depth_to_normals  # noqa: E402

from golden_cases import all_cases, t8_batches, H, W  # noqa: E402


def camera(f):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = W / 2.0
    K[:, 1, 2] = H / 2.0
    return K


def normals_for(depth, prm):
    n = depth_to_normals(torch.from_numpy(depth)[:, None] + prm["normal_z_offset"], camera(prm["focal"]))
    n[:, 1] = -n[:, 1]
    return n


CASES = list(all_cases())


@pytest.mark.parametrize("name,case", CASES, ids=[c[0] for c in CASES])
def test_c_oracle_matches_reference_golden(name, case):
    prm, exp = case["params"], case["expect"]
    unit, pt = c_oracle.light_prep(case["light"], clamp_z_min=prm["clamp_light_z_min"],
                                   light_distance=prm["light_distance"])
    np.testing.assert_array_equal(unit, exp["unit_light_direction"].reshape(-1, 3))
    tt = c_oracle.sample_table(prm["t0"], prm["dt"], prm["n_samples"])
    md, _ = c_oracle.shadow_min_distance(case["depth"], case["mask"], pt[:, None, :], tt,
                                         bonus=prm["bonus"], bonus_box=prm["bonus_box"])
    nrm = normals_for(case["depth"], prm).numpy()
    o = c_oracle.shade(nrm, case["depth"], case["albedo"], pt[:, None, :], case["ambient"][:, None], md,
                       intensity=prm["intensity"])
    if "shadow_mask_weights" in exp:
        assert np.abs(o["shadow_w"][:, 0] - exp["shadow_mask_weights"]).max() <= 2e-6
    # minimum_distance and argmin as the reference's own torch.min returned them (T8:514), before the +5 bonus
    md_raw, am = c_oracle.shadow_min_distance(case["depth"], case["mask"], pt[:, None, :], tt)
    md_ref, am_ref = exp["minimum_distance"], exp["argmin"].astype(np.int32)
    lit = md_ref < 1e5
    assert np.array_equal(lit, md_raw[:, 0] < 1e5)                       # same pixels end on a masked minimum
    np.testing.assert_array_equal(md_raw[:, 0][~lit], md_ref[~lit])      # ... with the reference's value 1e6
    err = np.abs(md_raw[:, 0][lit] - md_ref[lit])
    assert err.max() <= 1e-5 * max(1.0, float(md_ref[lit].max()))        # residue: torch-CPU's vectorised sqrt (1 ulp)
    assert (err == 0).mean() >= 0.98
    # argmin: identical wherever the oracle's distance equals the reference's bit for bit; a 1-ulp sqrt
    # difference can move the first index inside a run of near-tied samples (never observed to exceed 0.1 %)
    same_bits = lit & (md_raw[:, 0] == md_ref)
    assert np.array_equal(am[:, 0][same_bits], am_ref[same_bits])
    assert (am[:, 0][lit] == am_ref[lit]).mean() >= 0.999
    if "full_shading" in exp:
        assert np.abs(o["full_shading"][:, 0] - exp["full_shading"]).max() <= 1e-6
    if "rendered_images" in exp:
        assert np.abs(o["rendered"][:, 0] - exp["rendered_images"]).max() <= 1e-6
    if "rendered_images_face0" in exp:
        assert np.abs(o["rendered"][:1, 0] - exp["rendered_images_face0"]).max() <= 1e-6


def test_sample_table_is_numpy_arange():
    np.testing.assert_array_equal(c_oracle.sample_table(0.025, 0.005, 160), np.arange(0.025, 0.825, 0.005))
    np.testing.assert_array_equal(c_oracle.sample_table(0.03, 0.005, 159), np.arange(0.03, 0.825, 0.005))
    np.testing.assert_array_equal(M.BlockParams().sample_table(), np.arange(0.025, 0.825, 0.005))
    assert len(np.arange(0.025, 0.825, 0.0025)) == 320
    np.testing.assert_array_equal(c_oracle.sample_table(0.025, 0.0025, 320), np.arange(0.025, 0.825, 0.0025))


def test_materialised_port_matches_golden_forward_and_grads():
    """One T8 batch with autograd: forward bit-level, gradients against the reference's autograd."""
    name, case = next(t8_batches())
    assert name == "t8_a"
    exp = case["expect"]
    depth = torch.from_numpy(case["depth"])[:, None].clone().requires_grad_()
    alb = torch.from_numpy(case["albedo"]).clone().requires_grad_()
    light = torch.from_numpy(case["light"]).clone().requires_grad_()
    amb = torch.from_numpy(case["ambient"]).clone().requires_grad_()
    n = depth_to_normals(depth + 1610.0, camera(1570.0))
    n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], 1)
    o = M.render_block(depth, alb, light, amb, n, torch.from_numpy(case["mask"]))
    assert np.abs(o["shadow_mask_weights"].detach().numpy() - exp["shadow_mask_weights"]).max() <= 1e-7
    assert np.abs(o["rendered_images"].detach().numpy() - exp["rendered_images"]).max() <= 1e-7
    rng = np.random.default_rng(int(exp["grad_full_seed"]))
    G_r = torch.from_numpy(rng.random((3, 3, H, W), dtype=np.float32))
    G_w = torch.from_numpy(rng.random((3, H, W), dtype=np.float32))
    loss = (o["rendered_images"] * G_r).sum() + (o["shadow_mask_weights"] * G_w).sum()
    loss.backward()
    gd = exp["grad_full_depth"]
    assert np.abs(depth.grad[:, 0].numpy() - gd).max() <= 1e-5 * np.abs(gd).max()
    gl = exp["grad_full_light4"]
    np.testing.assert_allclose(light.grad.numpy(), gl[:, 1:4], rtol=1e-5, atol=1e-5 * np.abs(gl).max())
    np.testing.assert_allclose(amb.grad.numpy(), gl[:, 0], rtol=1e-6)
    ga = exp["grad_full_albedo"]
    assert np.abs(alb.grad[0].numpy() - ga).max() <= 1e-5


def test_c_oracle_equals_materialised_small():
    """The two oracles agree with each other at a size/N with no reference run (generalisation check)."""
    Hs, Ws, N = 48, 64, 37
    rng = np.random.default_rng(5)
    r, c = np.mgrid[0:Hs, 0:Ws]
    depth = (20 * np.exp(-(((c - 30) / 14.0) ** 2 + ((r - 22) / 11.0) ** 2)) + rng.random((Hs, Ws))).astype(np.float32)
    mask = (rng.random((Hs, Ws)) > 0.2).astype(np.uint8)
    lights = np.array([[0.3, 0.5, 0.8], [-0.9, 0.1, 0.2], [0.001, -0.002, 1.0], [0.7, -0.7, 0.05]], np.float32)
    p = M.BlockParams(n_samples=N, t0=0.02, dt=0.02)
    tt = c_oracle.sample_table(0.02, 0.02, N)
    unit, pt = c_oracle.light_prep(lights, clamp_z_min=0.0)
    B = len(lights)
    md, am = c_oracle.shadow_min_distance(np.repeat(depth[None], B, 0), mask[None], pt[:, None, :], tt)
    for b in range(B):
        v, idx = M.min_distance_one(torch.from_numpy(depth), torch.from_numpy(mask), torch.from_numpy(pt[b]), p)
        assert np.abs(v.numpy() - md[b, 0]).max() <= 2e-5, b
        assert (idx.numpy() == am[b, 0]).mean() > 0.999
