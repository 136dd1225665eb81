"""GPU: the inference-side byte kernels (csrc/gcfr_postprocess.hip) against the host statements of the same script
lines (geomconsistentfr_amd/postprocess.py: composite_into_input, diagnostic_images, to_uint8, fix_border_artifacts),
byte for byte -- half-way cases included -- and against scipy for the MATLAB border fix."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")


def _inputs(B, H, W, seed):
    rng = np.random.default_rng(seed)
    f = lambda *s: rng.random(s, dtype=np.float32)
    x = f(B, H, W, 3)
    ren, alb = f(B, 3, H, W) * 1.2 - 0.1, f(B, 3, H, W)                       # a few values beyond [0, 1]: saturation
    depth = (60 * rng.standard_normal((B, 1, H, W))).astype(np.float32)
    w, fin = f(B, H, W), f(B, H, W) * 1.1
    nrm = rng.standard_normal((B, 3, H, W)).astype(np.float32)
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    mask_u8 = rng.choice([0, 64, 128, 192, 255], size=(H, W), p=[0.4, 0.1, 0.1, 0.1, 0.3]).astype(np.uint8)
    # plant exact half-way cases: 255*r*m == k + 0.5
    ren[0, 0, 0, :8] = np.float32((np.arange(8) + 0.5) / 255.0)
    mask_u8[0, :8] = 255
    return x, ren, alb, depth, w, fin, nrm, mask_u8


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (3, 64, 96), (2, 37, 51)])
def test_inference_images_match_the_host_statement_byte_for_byte(B, H, W):
    from geomconsistentfr_amd import postprocess as pp
    x, ren, alb, depth, w, fin, nrm, mask_u8 = _inputs(B, H, W, B * 1000 + W)
    m01 = mask_u8 / 255.0                                                     # S1:580: uint8 / python float -> f64
    t = lambda a: torch.from_numpy(a).to(DEV)
    got = pp.inference_images_device(t(x), t(ren), t(mask_u8), albedo=t(alb), depth=t(depth), shadow_mask_weights=t(w),
                                     final_shading=t(fin), surface_normals=t(nrm))
    torch.cuda.synchronize()
    for b in range(B):
        exp = pp.diagnostic_images(x[b].astype(np.float64), alb[b], depth, b, w[b], ren[b], fin[b], nrm[b],
                                   m01)
        for k, v in exp.items():
            np.testing.assert_array_equal(got[k][b].cpu().numpy(), pp.to_uint8(v), err_msg="%s face %d" % (k, b))
    # only the composite, per-face masks
    masks = np.stack([np.roll(mask_u8, 3 * b, axis=1) for b in range(B)])
    only = pp.inference_images_device(t(x), t(ren), t(masks))
    assert set(only) == {"rendered_image"}
    for b in range(B):
        np.testing.assert_array_equal(only["rendered_image"][b].cpu().numpy(),
                                      pp.to_uint8(pp.composite_into_input(x[b].astype(np.float64), ren[b], masks[b] / 255.0)))


@pytest.mark.parametrize("B,H,W", [(1, 256, 256), (2, 40, 36), (2, 33, 70)])
def test_border_fix_matches_matlab_semantics(B, H, W):
    import scipy.ndimage as ndi
    from geomconsistentfr_amd import postprocess as pp
    rng = np.random.default_rng(H)
    img = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    mask = np.zeros((B, H, W), np.uint8)
    mask[:, H // 5:H - H // 4, W // 6:W - W // 5] = rng.choice([64, 127, 128, 255], size=(B, H - H // 4 - H // 5, W - W // 5 - W // 6))
    mask[0, :3, :] = 255                                                       # a mask touching the image border: zero padding
    got = pp.fix_border_artifacts_device(torch.from_numpy(img).to(DEV), torch.from_numpy(mask).to(DEV)).cpu().numpy()
    n_border = 0
    for b in range(B):
        np.testing.assert_array_equal(got[b], pp.fix_border_artifacts(img[b], mask[b]))          # host statement
        m = np.floor(mask[b] / 255.0 + 0.5)                                                        # MATLAB uint8 division
        conv = ndi.correlate(m, np.ones((7, 7)), mode="constant", cval=0.0)
        border = (conv < 30) & (conv > 0)
        n_border += int(border.sum())
        for c in range(3):                                                                          # scipy, independently
            med = ndi.median_filter(img[b, ..., c], size=3, mode="constant", cval=0)
            np.testing.assert_array_equal(got[b, ..., c][border], med[border])
        np.testing.assert_array_equal(got[b][~border], img[b][~border])
    assert n_border > 0
    # one mask shared by the batch
    shared = pp.fix_border_artifacts_device(torch.from_numpy(img).to(DEV), torch.from_numpy(mask[B - 1]).to(DEV)).cpu().numpy()
    np.testing.assert_array_equal(shared[0], pp.fix_border_artifacts(img[0], mask[B - 1]))


def test_device_functions_have_no_cpu_path():
    from geomconsistentfr_amd import postprocess as pp
    from geomconsistentfr_amd._lib import GcfrError
    with pytest.raises(GcfrError):
        pp.inference_images_device(torch.zeros(1, 8, 8, 3), torch.zeros(1, 3, 8, 8), torch.ones(1, 8, 8, dtype=torch.uint8))
    with pytest.raises(GcfrError):
        pp.fix_border_artifacts_device(torch.zeros(1, 8, 8, 3, dtype=torch.uint8), torch.ones(8, 8, dtype=torch.uint8))
