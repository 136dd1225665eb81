"""GPU: the outer drop-in boundary -- RelightNet.forward mirrors (T8:196/524, S1:169/505, SLT:169/514)
against the golden tuples produced by the reference's own forward, by injecting the same head outputs."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from golden_cases import GOLDEN, t8_batches, _inputs, H, W  # noqa: E402

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


class Const(torch.nn.Module):
    def __init__(self, v):
        super().__init__()
        self.v = v

    def forward(self, _):
        return self.v


def logit(a):
    a = np.clip(a.astype(np.float64), 1e-6, 1 - 1e-6)
    return np.log(a / (1 - a)).astype(np.float32)


def camera(f):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = W / 2.0
    K[:, 1, 2] = H / 2.0
    return K.to(DEV)


def inject(net, depth, albedo, raw4):
    B = depth.shape[0]
    net.conv_depth_c2_o = Const(torch.from_numpy(depth / np.float32(100.0))[:, None].to(DEV))
    net.conv_albedo_c2_o = Const(torch.from_numpy(logit(albedo)).to(DEV))
    net.linear_SL2 = Const(torch.from_numpy(raw4.astype(np.float32)).view(B, 1, 1, 4).to(DEV))
    return net.to(DEV).eval()


@pytest.mark.parametrize("idx", [0, 2, 6])          # 6 = t8_g: noise-40 / noise-400 / untrained-network depth, (1,0,0), (0,0,1), z < 0
def test_training_forward_returns_the_reference_8_tuple(idx):
    from geomconsistentfr_amd.relightnet import RelightNet
    name, case = list(t8_batches())[idx]
    exp = case["expect"]
    raw4 = np.concatenate([case["ambient"][:, None], case["light"]], 1)
    net = inject(RelightNet(), case["depth"], case["albedo"], raw4)
    masks = torch.from_numpy(case["mask"].astype(np.float64))[..., None].to(DEV)        # (B,H,W,1) f64 as T8:612
    with torch.no_grad():
        out = net(torch.zeros(3, H, W, 3, device=DEV), 200, camera(1570.0), masks)
    assert len(out) == 8
    shapes = [(3, 3, H, W), (3, 1, H, W), (3, H, W), (3, H, W), (3, H, W), (3, 3, H, W), (3, 3, 1, 1), (3, 1, 1)]
    assert [tuple(o.shape) for o in out] == shapes
    assert np.abs(out[0].cpu().numpy() - case["albedo"]).max() <= 1e-6
    np.testing.assert_array_equal(out[1].cpu().numpy()[:, 0], case["depth"])
    assert np.abs(out[2].cpu().numpy() - exp["shadow_mask_weights"]).max() <= 1e-4
    assert np.abs(out[4].cpu().numpy() - exp["full_shading"]).max() <= 1e-5
    if "rendered_images" in exp:
        assert np.abs(out[5].cpu().numpy() - exp["rendered_images"]).max() <= 1e-3
    np.testing.assert_allclose(out[6].cpu().numpy().reshape(3, 3), exp["unit_light_direction"], atol=1e-7)
    np.testing.assert_array_equal(out[7].cpu().numpy().reshape(3), exp["ambient_values"])
    np.testing.assert_array_equal(out[3].cpu().numpy()[:, 0, 0], exp["ambient_values"])


@pytest.mark.parametrize("name", ["s1_a", "s1_e", "slt_a", "slt_b"])
def test_inference_forwards_return_the_reference_tuples(name):
    from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer, RelightNetSingleImage
    depths, masks, alb = _inputs()
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    depth, mask = depths[int(z["depth_idx"])][None], masks[int(z["mask_idx"])]
    s1 = name.startswith("s1")
    net = inject(RelightNetSingleImage() if s1 else RelightNetLightingTransfer(), depth, alb[None], z["raw4"][None])
    m = torch.from_numpy(mask.astype(np.float64))[..., None].to(DEV)                    # (H,W,1) as S1:580
    tl = torch.from_numpy(z["target_light"]).view(1, 3, 1, 1).to(DEV)
    ta = torch.tensor([float(z["target_ambient"]) if not s1 else 0.0], device=DEV).view(1, 1, 1)
    with torch.no_grad():
        if s1:
            out = net(torch.zeros(1, H, W, 3, device=DEV), 200, camera(1570.0), m, tl, ta, m[None])
        else:
            out = net(torch.zeros(1, H, W, 3, device=DEV), 200, camera(700.0), m, tl, ta)
    assert len(out) == (10 if s1 else 12)
    assert np.abs(out[2].cpu().numpy()[0] - z["shadow_mask_weights"]).max() <= 1e-4
    assert np.abs(out[4].cpu().numpy()[0] - z["full_shading"]).max() <= 1e-5
    if "rendered_images" in z.files:
        assert np.abs(out[5].cpu().numpy()[0] - z["rendered_images"]).max() <= 1e-3
    np.testing.assert_allclose(out[6].cpu().numpy().reshape(3), z["unit_light_direction"], atol=1e-7)
    np.testing.assert_allclose(out[7].cpu().numpy().reshape(1), z["ambient_values"], atol=1e-7)
    assert tuple(out[8].shape) == (1, H, W) and tuple(out[9].shape) == (1, 3, H, W)


def test_training_step_backward_reaches_every_parameter():
    from geomconsistentfr_amd.relightnet import RelightNet
    torch.manual_seed(0)
    net = RelightNet().to(DEV).train()
    B = 4                                                                               # not the reference's 3
    img = torch.rand(B, H, W, 3, device=DEV)
    masks = torch.ones(B, H, W, 1, dtype=torch.float64, device=DEV)
    out = net(img, 200, camera(1570.0), masks)
    loss = out[5].mean() + out[2].mean() + out[6].sum() + out[7].sum() + out[1].abs().mean() * 1e-3
    loss.backward()
    missing = [n for n, p in net.named_parameters() if p.grad is None]
    assert not missing, missing
    assert all(torch.isfinite(p.grad).all() for p in net.parameters())


def test_trainer_steps_run_and_update_parameters():
    """Full training step (T8:617-656) around the HIP block on one GPU, batch 4, two iterations."""
    from geomconsistentfr_amd.train import TrainConfig, Trainer, synthetic_batch
    torch.manual_seed(0)
    tr = Trainer(TrainConfig(miopen_find=False), device=DEV)      # find mode costs ~50 s of tuning on first use
    before = torch.cat([p.detach().flatten().clone() for p in tr.model.parameters()])
    d_before = torch.cat([p.detach().flatten().clone() for p in tr.patchgan.parameters()])
    batch = synthetic_batch(4, 0, device=DEV)
    logs0 = tr.step(batch, epoch=200, j=0)
    logs1 = tr.step(batch, epoch=200, j=1)
    assert "discriminator" in logs0 and "discriminator" not in logs1          # D step every GD_ratio (T8:624)
    for k in ("recon", "depth", "ambient", "lighting", "albedo", "generator", "DSSIM", "total"):
        assert np.isfinite(logs0[k]) and np.isfinite(logs1[k]), k
    after = torch.cat([p.detach().flatten() for p in tr.model.parameters()])
    d_after = torch.cat([p.detach().flatten() for p in tr.patchgan.parameters()])
    assert not torch.equal(before, after) and not torch.equal(d_before, d_after)
    assert torch.isfinite(after).all()


def test_inference_helpers_follow_the_scripts():
    """relight_single_image / relight_batch / lighting_transfer == the manual S1 / SLT call sequences."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import postprocess_statements as pp            # checker only
    from geomconsistentfr_amd.inference import (LIGHT_DIRECTIONS, camera_matrix, lighting_transfer, relight_batch,
                                                relight_single_image)
    from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer, RelightNetSingleImage
    assert len(LIGHT_DIRECTIONS) == 11
    torch.manual_seed(0)
    rng = np.random.default_rng(0)
    img = rng.random((H, W, 3)).astype(np.float32)
    ref = rng.random((H, W, 3)).astype(np.float32)
    r, c = np.mgrid[0:H, 0:W]
    mask = np.where((((c - 128) / 80.0) ** 2 + ((r - 128) / 100.0) ** 2) < 1, 255, 0).astype(np.uint8)
    mask[100:110, 100:110] = 64                                              # the shipped masks have 4 grey levels

    s1 = RelightNetSingleImage().to(DEV).eval()
    light = LIGHT_DIRECTIONS["top_A00E45"]
    comp = relight_single_image(s1, img, mask, light, device=DEV)
    assert comp.dtype == np.uint8 and comp.shape == (H, W, 3)
    with torch.no_grad():                                                    # the manual sequence of S1:582-620
        m = torch.from_numpy(mask.astype(np.float64)).reshape(H, W, 1).to(DEV) / 255.0
        out = s1(torch.from_numpy(img)[None].to(DEV), 200, camera_matrix(1570.0, H, W, DEV), m,
                 torch.tensor(light, dtype=torch.float32, device=DEV).view(1, 3, 1, 1),
                 torch.full((1, 1, 1), 0.5, device=DEV), m[None])
    exp = pp.to_uint8(pp.composite_into_input(img.astype(np.float64), out[5][0].cpu().numpy(), mask / 255.0))
    assert np.abs(comp.astype(int) - exp.astype(int)).max() <= 1      # two forwards: MIOpen may pick different conv paths
    np.testing.assert_array_equal(comp[mask == 0], pp.to_uint8(img.astype(np.float64) * 255.0)[mask == 0])

    lights = np.array([LIGHT_DIRECTIONS[k] for k in ("multipie_04", "multipie_18", "bottom_left_A60E-20")], np.float32)
    outs = relight_batch(s1, np.stack([img, ref, img]), mask, lights, device=DEV)
    assert len(outs) == 10 and tuple(outs[5].shape) == (3, 3, H, W)
    single = relight_batch(s1, img[None], mask, lights[:1], device=DEV)
    assert float((outs[2][0] - single[2][0]).abs().max()) <= 1e-5           # batch item == single forward (eval BN)

    slt = RelightNetLightingTransfer().to(DEV).eval()
    res = lighting_transfer(slt, img, ref, mask, device=DEV)
    assert {"rendered_image", "shadow_mask", "albedo", "depth", "shading", "surface_normals", "estimated_light",
            "estimated_ambient"} <= set(res)
    assert abs(float(np.linalg.norm(res["estimated_light"])) - 1.0) < 1e-5 and res["estimated_light"][2] > 0
