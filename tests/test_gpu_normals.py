"""GPU: gcfr_normals_fwd / gcfr_normals_bwd against the oracle's kornia restatement (torch CPU, f64) and
its autograd.  NB: this stage is parity-UNPINNED with respect to kornia itself (un-vendored dependency)."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def camera(f, H, W):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    return K


@pytest.mark.parametrize("H,W,f,off", [(64, 80, 1570.0, 1610.0), (256, 256, 700.0, 1410.0), (31, 17, 300.0, 50.0)])
def test_normals_forward_and_backward_match_the_restatement(H, W, f, off):
    from geomconsistentfr_amd.normals import depth_to_normals
    from normals_restatement import depth_to_normals as oracle_normals
    rng = np.random.default_rng(H * W)
    depth = (25 * rng.random((3, 1, H, W))).astype(np.float32)
    G = rng.standard_normal((3, 3, H, W)).astype(np.float32)
    K = camera(f, H, W)
    d_ref = torch.from_numpy(depth).clone().requires_grad_()
    n_ref = oracle_normals(d_ref + off, K)
    n_ref = torch.cat([n_ref[:, 0:1], -n_ref[:, 1:2], n_ref[:, 2:3]], 1)
    (n_ref * torch.from_numpy(G)).sum().backward()

    d = torch.from_numpy(depth).to(DEV).requires_grad_()
    n = depth_to_normals(d, K.to(DEV), z_offset=off)
    assert n.dtype == torch.float32 and n.shape == (3, 3, H, W)
    assert float((n.detach().cpu().double() - n_ref.detach()).abs().max()) <= 2e-6
    (n * torch.from_numpy(G).to(DEV)).sum().backward()
    g, g_ref = d.grad.cpu().numpy(), d_ref.grad.numpy()
    assert np.abs(g - g_ref).max() <= 2e-5 * max(np.abs(g_ref).max(), 1e-6)


def test_normals_per_image_camera_matrices():
    from geomconsistentfr_amd.normals import depth_to_normals
    rng = np.random.default_rng(1)
    depth = torch.from_numpy((25 * rng.random((2, 1, 32, 32))).astype(np.float32)).to(DEV)
    K = torch.cat([camera(1570.0, 32, 32), camera(700.0, 32, 32)]).to(DEV)
    n = depth_to_normals(depth, K, z_offset=100.0)
    n0 = depth_to_normals(depth[:1], K[:1], z_offset=100.0)
    n1 = depth_to_normals(depth[1:], K[1:], z_offset=100.0)
    assert torch.equal(n, torch.cat([n0, n1]))
    np.testing.assert_allclose(n.norm(dim=1).cpu().numpy(), 1.0, atol=1e-6)


def test_fused_normals_forward_is_bit_identical_to_the_three_stage_path():
    """gcfr_render_from_depth_fwd (normals stencil inside the march epilogue) == gcfr_normals_fwd followed by
    gcfr_render_fwd, bit for bit, including the returned normals; and render_from_depth picks it under no_grad."""
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    from geomconsistentfr_amd.normals import depth_to_normals
    rng = np.random.default_rng(11)
    B, H, W = 3, 96, 128
    dev = torch.device(DEV)
    depth = torch.from_numpy((30 * rng.random((B, H, W))).astype(np.float32)).to(dev)
    mask = torch.from_numpy((rng.random((B, H, W)) > 0.3).astype(np.uint8)).to(dev)
    albedo = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).to(dev)
    light = torch.from_numpy(rng.standard_normal((B, 2, 3)).astype(np.float32)).to(dev)
    amb = torch.from_numpy(rng.random((B, 2), dtype=np.float32)).to(dev)
    K = camera(1570.0, H, W).to(dev)
    prm = RenderParams(n_samples=64, dt=0.0125)
    n = depth_to_normals(depth[:, None], K, z_offset=1610.0)
    a = R.render_fwd(depth, mask, light, amb, n, albedo, prm, want_argmin=True)
    b = R.render_fwd(depth, mask, light, amb, None, albedo, prm, want_argmin=True,
                     camera=(1570.0, 1570.0, W / 2.0, H / 2.0, 1610.0))
    for k in ("minimum_distance", "argmin", "shadow_mask_weights", "full_shading", "final_shading", "rendered_images"):
        assert torch.equal(a[k], b[k]), k
    assert torch.equal(b["surface_normals"], n)
    with torch.no_grad():
        r = R.render_from_depth(depth[:, None], albedo, light[:, 0], amb[:, 0], K, 1610.0, mask, prm)
    assert torch.equal(r["rendered_images"], a["rendered_images"][:, 0])
    assert torch.equal(r["surface_normals"], n)


def test_many_lights_per_face_run_the_normals_stage_as_its_own_launch_with_the_same_bits(monkeypatch):
    """block.normals_stage_for: from NORMALS_KERNEL_MIN_LIGHTS lights per face on, render_fwd(camera=...) and RenderFwdPlan run
    gcfr_normals_fwd once and hand its output to gcfr_render_fwd, instead of evaluating the light-independent stencil in every
    light's epilogue (config 5: 18 lights per face).  Every output equals the fused form's bit for bit; a hipGraph of the plan
    and the two-phase form (prepass hoisted) as well."""
    from geomconsistentfr_amd import RenderParams
    from geomconsistentfr_amd import block as R
    rng = np.random.default_rng(12)
    B, L, H, W = 2, 17, 96, 128
    assert L >= R.NORMALS_KERNEL_MIN_LIGHTS and R.normals_stage_for(L) == "kernel" and R.normals_stage_for(1) == "fused"
    dev = torch.device(DEV)
    depth = torch.from_numpy((30 * rng.random((B, H, W))).astype(np.float32)).to(dev)
    mask = torch.from_numpy((rng.random((B, H, W)) > 0.3).astype(np.uint8)).to(dev)
    albedo = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).to(dev)
    light = torch.from_numpy(rng.standard_normal((B, L, 3)).astype(np.float32)).to(dev)
    amb = torch.from_numpy(rng.random((B, L), dtype=np.float32)).to(dev)
    cam = (1570.0, 1570.0, W / 2.0, H / 2.0, 1610.0)
    prm = RenderParams(n_samples=48, dt=0.016)
    keys = ("minimum_distance", "argmin", "shadow_mask_weights", "full_shading", "final_shading", "rendered_images", "surface_normals")
    own = R.render_fwd(depth, mask, light, amb, None, albedo, prm, want_argmin=True, camera=cam)
    pre = R.render_prepass(depth, mask, light, prm, want_argmin=True)
    own2 = R.render_fwd(depth, mask, light, amb, None, albedo, prm, want_argmin=True, camera=cam, prepared=pre)
    plan = R.RenderFwdPlan(B, L, H, W, prm, dev, want_argmin=True, camera=cam)
    assert plan.normals_stage == "kernel"
    args = (depth, mask, light, amb, None, albedo)
    planned = {k: v.clone() for k, v in plan(*args).items() if k in keys}
    graphed = {k: v.clone() for k, v in R.RenderFwdPlan(B, L, H, W, prm, dev, want_argmin=True, camera=cam).capture(*args).replay().items()
               if k in keys}
    monkeypatch.setattr(R, "NORMALS_KERNEL_MIN_LIGHTS", 1 << 30)
    assert R.normals_stage_for(L) == "fused"
    fused = R.render_fwd(depth, mask, light, amb, None, albedo, prm, want_argmin=True, camera=cam)
    assert R.RenderFwdPlan(B, L, H, W, prm, dev, want_argmin=True, camera=cam).normals_stage == "fused"
    for k in keys:
        for name, got in (("own launch", own), ("own launch, prepass hoisted", own2), ("plan", planned), ("hipGraph", graphed)):
            assert torch.equal(got[k], fused[k]), (name, k)


def test_forward_normals_tolerance_contract_and_non_finite_cells():
    """include/gcfr.h: forward normals are NOT bit-pinned (kornia's summation order is unspecified and the kernel uses
    reciprocals / rsq + Newton steps instead of IEEE divisions, round-2 advisor note): the contract is `within 4 f32 ulp of
    the f64 restatement after rounding to f32`, and a non-finite depth cell poisons exactly the normals whose 3 x 3
    neighbourhood (clamped at the border) contains it -- the same pixels as in the restatement."""
    from geomconsistentfr_amd.normals import depth_to_normals
    from normals_restatement import depth_to_normals as oracle_normals
    H, W, f, off = 96, 128, 1570.0, 1610.0
    r, c = np.mgrid[0:H, 0:W]
    depth = (60 * np.exp(-(((c - 64) / 40.0) ** 2 + ((r - 48) / 30.0) ** 2)) + 2 * np.sin(c / 5.0) * np.cos(r / 7.0)).astype(np.float32)
    depth = np.stack([depth, depth[::-1].copy()])[:, None]
    K = camera(f, H, W)
    n_ref = oracle_normals(torch.from_numpy(depth) + off, K)
    n_ref = torch.cat([n_ref[:, 0:1], -n_ref[:, 1:2], n_ref[:, 2:3]], 1).numpy()
    n = depth_to_normals(torch.from_numpy(depth).to(DEV), K.to(DEV), z_offset=off).cpu().numpy()
    ulp = np.spacing(np.maximum(np.abs(n_ref), 2.0 ** -10).astype(np.float32))
    assert (np.abs(n.astype(np.float64) - n_ref) <= 4 * ulp).all()
    # non-finite cells: interior NaN, interior +inf, a NaN in the corner (replicate padding folds three offsets onto it)
    bad = depth.copy()
    bad[0, 0, 40, 50] = np.nan
    bad[0, 0, 70, 20] = np.inf
    bad[1, 0, 0, 0] = np.nan
    nb_ref = oracle_normals(torch.from_numpy(bad) + off, K).numpy()
    nb = depth_to_normals(torch.from_numpy(bad).to(DEV), K.to(DEV), z_offset=off).cpu().numpy()
    poisoned_ref = ~np.isfinite(nb_ref).all(axis=1)
    poisoned = ~np.isfinite(nb).all(axis=1)
    assert poisoned_ref.sum() == 9 + 9 + 4
    np.testing.assert_array_equal(poisoned, poisoned_ref)
    ok = ~poisoned
    assert (np.abs(nb.astype(np.float64) - np.concatenate([nb_ref[:, 0:1], -nb_ref[:, 1:2], nb_ref[:, 2:3]], 1))[np.broadcast_to(ok[:, None], nb.shape)] <= 1e-6).all()
