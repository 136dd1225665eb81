"""CPU: the batch-agnostic RelightNet mirror -- structure, checkpoint compatibility, normals restatement.
(The render block itself has no CPU path; these tests stop at the T8:352 seam.)"""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
REF = "/root/reference"
needs_ref = pytest.mark.skipif(not os.path.isdir(REF), reason="reference tree only exists in the authoring container")


def test_parameter_counts_match_the_reference():
    from geomconsistentfr_amd.relightnet import PatchGAN, RelightNet
    n = lambda m: sum(p.numel() for p in m.parameters())
    assert n(RelightNet("3x3")) == 1_204_796          # SURVEY.md section 2 [probe] of T8's RelightNet
    assert n(RelightNet("1x1")) == 932_449            # train_lighting_transfer.py variant
    assert n(PatchGAN()) == 2_766_529


def test_features_are_batch_agnostic_and_shaped_like_the_reference():
    from geomconsistentfr_amd.relightnet import PatchGAN, RelightNet
    torch.manual_seed(0)
    net = RelightNet().eval()
    for B in (1, 2, 5):
        with torch.no_grad():
            albedo, depth, SL = net.features(torch.rand(B, 256, 256, 3), epoch=200)
        assert albedo.shape == (B, 3, 256, 256) and depth.shape == (B, 1, 256, 256) and SL.shape == (B, 1, 1, 4)
        assert float(albedo.min()) > 0 and float(albedo.max()) < 1
    assert PatchGAN()(torch.rand(2, 3, 256, 256)).shape == (2, 1, 15, 15)


def test_epoch_gates_switch_the_skip_connections():
    from geomconsistentfr_amd.relightnet import RelightNet
    torch.manual_seed(0)
    net = RelightNet().eval()
    x = torch.rand(1, 256, 256, 3)
    with torch.no_grad():
        outs = [net.features(x, e)[1] for e in (8, 9, 11, 13, 15, 200)]
    for a, b in zip(outs[:-2], outs[1:-1]):
        assert not torch.equal(a, b)                  # gates at epoch > 8, 10, 12, 14 (T8:245-283)
    assert torch.equal(outs[-2], outs[-1])


def test_product_normals_refuse_host_tensors():
    from geomconsistentfr_amd._lib import GcfrError
    from geomconsistentfr_amd.normals import depth_to_normals
    with pytest.raises(GcfrError):
        depth_to_normals(torch.zeros(1, 1, 8, 8), torch.eye(3, dtype=torch.float64)[None])


def test_render_plan_and_render_block_refuse_the_host():
    """No CPU fallback anywhere on the product path: host tensors / host devices raise, they are never routed
    to the oracle."""
    from geomconsistentfr_amd._lib import GcfrError
    from geomconsistentfr_amd import block as R
    with pytest.raises(GcfrError):
        R.RenderFwdPlan(1, 1, 8, 8, device="cpu")
    z = torch.zeros(1, 8, 8)
    with pytest.raises(GcfrError):
        R.render_fwd(z, torch.ones(1, 8, 8), torch.ones(1, 1, 3), torch.ones(1, 1), torch.zeros(1, 3, 8, 8),
                     torch.zeros(1, 3, 8, 8))


@needs_ref
def test_reference_checkpoint_loads_and_network_outputs_match_the_reference():
    """The shipped lighting-transfer checkpoint loads with strict=True, and albedo / depth / light head
    equal the reference network's on the same image (CPU, eval mode)."""
    import ref_shim
    from PIL import Image
    from geomconsistentfr_amd.relightnet import RelightNetLightingTransfer
    sd = torch.load(os.path.join(REF, "model_lighting_transfer", "model_epoch106.pth"), map_location="cpu")
    mine = RelightNetLightingTransfer()
    mine.load_state_dict(sd, strict=True)
    mine.eval()
    SLT = ref_shim.load("SLT")
    ref = SLT.RelightNet()
    ref.load_state_dict(sd)
    ref = ref.float().eval()
    img = Image.open(os.path.join(REF, "sample_test_images_FFHQ", "00295.png")).convert("RGB").resize((256, 256))
    x = torch.from_numpy(np.asarray(img, dtype=np.float32) / 255.0)[None]
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = 700.0
    K[:, 2, 2] = 1.0
    K[:, 0, 2] = K[:, 1, 2] = 128.0
    with torch.no_grad():
        albedo, depth, SL = mine.features(x, 200)
        out = ref(x, 200, K, torch.ones(256, 256, 1, dtype=torch.float64),
                  torch.tensor([0.0, 0.7, 0.7]).view(1, 3, 1, 1), torch.tensor([0.5]).view(1, 1, 1))
    assert float((albedo - out[0]).abs().max()) <= 1e-5
    assert float((depth - out[1]).abs().max()) <= 1e-3          # depth is x100
    assert float((SL[:, :, :, 0] - out[11]).abs().max()) <= 1e-5
