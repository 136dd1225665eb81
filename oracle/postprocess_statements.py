"""TEST INFRASTRUCTURE -- numpy statements of what the reference's inference / evaluation scripts do with the render
block's outputs.  Only tests/ may import this module; the product's device kernels
(csrc/gcfr_postprocess.hip, csrc/gcfr_dataset.hip) are CHECKED against it, never routed through it.

Pinned to the reference itself: oracle/make_golden_slt_main.py runs the unmodified main() of
test_relight_single_image_lighting_transfer.py (SLT:516-579) and stores the relighting pass's model outputs next to the
six arrays the script hands to cv2.imwrite; tests/test_oracle_postprocess.py requires these statements to reproduce
those arrays BIT FOR BIT from those model outputs (and `to_uint8` to give the stored bytes).  Arithmetic keeps the
dtype numpy keeps in the scripts: `255.0 * a` stays f32 for an f32 array and f64 for an f64 one (the reference's
final_shading / surface_normals are f64 by promotion, its rendered images / albedo / shadow weights / depth f32), and
everything is widened to f64 by the f64 mask (`imread(mask)/255.0`, SLT:540 / S1:580).

  composite_into_input     test_relight_single_image.py:614-620 (S1) / S8:596-602 / SLT:567-572
  diagnostic_images        S8:583-608 / SLT:547-579: the six images per face
  to_uint8                 what cv2.imwrite does to a float image (saturate_cast<uchar>(cvRound(v)))
  fix_border_artifacts     fix_border_artifacts_CVPR2022.m:1-18 (3x3 median on the 7x7-box mask border)
  masked_mse               MSE_MP.m:24
  masked_dssim             DSSIM_MP_RGB.m:24-26 (MATLAB ssim: PARITY UNPINNED, MATLAB is not available)

Images are RGB, HWC; the reference's BGR flips exist only because it writes through cv2 and are not reproduced
(the fixtures' captured arrays are BGR and are flipped back by the test).
"""
from typing import Dict

import numpy as np


def to_uint8(img: np.ndarray) -> np.ndarray:
    """cv2.imwrite on a float array: saturate_cast<uchar>(cvRound(v)) -- round half to even, clip to [0,255]."""
    return np.clip(np.rint(np.asarray(img, dtype=np.float64)), 0, 255).astype(np.uint8)


def _mask3(mask: np.ndarray) -> np.ndarray:
    m = np.asarray(mask, dtype=np.float64)
    if m.ndim == 3:
        m = m[..., 0]
    return np.repeat(m[..., None], 3, axis=2)


def composite_into_input(input_image: np.ndarray, rendered: np.ndarray, mask: np.ndarray) -> np.ndarray:
    """Paste the relit face into the input photograph (S1:614-620, SLT:567-572).
    input_image (H,W,3) in [0,1] (f64 in the scripts: imread/255.0); rendered (3,H,W) (one item of rendered_images,
    f32); mask (H,W) in [0,1] (the reference divides the skin mask by 255: f64).  Returns (H,W,3) f64 in [0,255]."""
    m3 = _mask3(mask)
    out = np.asarray(input_image, dtype=np.float64) * 255.0
    # 255.0*rendered_images[k] keeps the array's dtype (f32 in the scripts), then the f64 3-channel mask widens the product
    ren = 255.0 * np.transpose(np.asarray(rendered), (1, 2, 0)) * m3
    sel = m3 > 0
    out[sel] = ren[sel]
    return out


def diagnostic_images(input_image, albedo, depth_batch, index, shadow_mask_weights, rendered, final_shading,
                      surface_normals, mask) -> Dict[str, np.ndarray]:
    """The six images S8:603-608 / SLT:574-579 write per face ([0,255], RGB / single channel).
    depth_batch (B,1,H,W) is needed whole: the reference min-max normalises -depth over the BATCH (S8:589-590;
    SLT:553-554 with B = 1); the other arguments are item `index` of the forward's outputs in their native layouts and
    dtypes: albedo (3,H,W), shadow_mask_weights (H,W), rendered (3,H,W), final_shading (H,W), surface_normals (3,H,W).
    `mask` in the dtype the script holds it: f64 in S1 / S8 (numpy f64 array / 255.0, S8:569), f32 in SLT (a torch uint8
    tensor / 255.0, SLT:540).  The 3-channel mask is always an f64 array (np.zeros, S8:579 / SLT:562) filled with those
    values; the single-channel products use the mask as it is, so in SLT shadow mask and depth map stay f32."""
    m3 = _mask3(mask)
    m1 = np.asarray(mask)
    m1 = m1[..., 0] if m1.ndim == 3 else m1                       # np.reshape(curr_mask_fill_nose.numpy(), (H, W))
    d = -np.asarray(depth_batch)                                   # f32 in the scripts
    d = (d - np.amin(d)) / (np.amax(d) - np.amin(d))
    hwc = lambda a: np.transpose(np.asarray(a), (1, 2, 0))
    return {
        "rendered_image": composite_into_input(input_image, rendered, mask),
        "shadow_mask": 255.0 * np.asarray(shadow_mask_weights) * m1,
        "albedo": 255.0 * hwc(albedo) * m3,
        "depth": 255.0 * d[index, 0] * m1,
        "shading": 255.0 * np.asarray(final_shading) * m1,
        "surface_normals": (255.0 * (hwc(surface_normals) + 1.0) / 2.0) * m3,
    }


# ------------------------------------------------------------------------------------------------
# fix_border_artifacts_CVPR2022.m
# ------------------------------------------------------------------------------------------------
def _medfilt3x3_zero_pad(ch: np.ndarray) -> np.ndarray:
    """MATLAB medfilt2 default: 3x3 neighbourhood, zero padding."""
    p = np.pad(ch, 1, mode="constant")
    stack = np.stack([p[i:i + ch.shape[0], j:j + ch.shape[1]] for i in range(3) for j in range(3)])
    return np.sort(stack, axis=0)[4]


def fix_border_artifacts(img_u8: np.ndarray, face_mask_u8: np.ndarray) -> np.ndarray:
    """img_u8 (H,W,3) uint8, face_mask_u8 (H,W) uint8 skin mask.  MATLAB semantics kept:
    `imread(mask)/255.0` is UINT8 division (round to nearest: 64 -> 0, 128 -> 1, 255 -> 1), the 7x7 box sum
    uses zero padding, the border is 0 < sum < 30, and border pixels take the 3x3 median of the image."""
    img = np.asarray(img_u8, dtype=np.uint8).copy()
    m = np.floor(np.asarray(face_mask_u8, dtype=np.float64) / 255.0 + 0.5)        # uint8 rounding division
    p = np.pad(m, 3, mode="constant")
    H, W = m.shape
    conv = sum(p[i:i + H, j:j + W] for i in range(7) for j in range(7))
    border = (conv < 30) & (conv > 0)
    for c in range(3):
        f = _medfilt3x3_zero_pad(img[..., c])
        img[..., c][border] = f[border]
    return img


# ------------------------------------------------------------------------------------------------
# offline metrics
# ------------------------------------------------------------------------------------------------
def masked_mse(recon_u8: np.ndarray, gt_u8: np.ndarray, mask_u8: np.ndarray) -> float:
    """MSE_MP.m:24: sum |r*m - g*m|^2 / (3 * sum m), images and mask scaled by 1/255."""
    r = np.asarray(recon_u8, dtype=np.float64) / 255.0
    g = np.asarray(gt_u8, dtype=np.float64) / 255.0
    m = np.asarray(mask_u8, dtype=np.float64) / 255.0
    m3 = m[..., None]
    return float((np.abs(r * m3 - g * m3) ** 2).sum() / (3.0 * m.sum()))


def _gauss3d_replicate(x: np.ndarray, sigma: float = 1.5) -> np.ndarray:
    """Separable Gaussian over all three axes of an (H,W,3) array, radius ceil(3 sigma), replicate padding --
    MATLAB's ssim treats an M x N x 3 input as a 3-D volume."""
    r = int(np.ceil(3 * sigma))
    k = np.exp(-(np.arange(-r, r + 1) ** 2) / (2 * sigma ** 2))
    k /= k.sum()
    out = x
    for ax in range(3):
        pad = [(0, 0)] * 3
        pad[ax] = (r, r)
        p = np.pad(out, pad, mode="edge")
        out = sum(k[i] * np.take(p, np.arange(i, i + x.shape[ax]), axis=ax) for i in range(2 * r + 1))
    return out


def masked_dssim(recon_u8: np.ndarray, gt_u8: np.ndarray, mask_u8: np.ndarray) -> float:
    """DSSIM_MP_RGB.m:24-26: (1 - masked mean of MATLAB ssim's map) / 2.  UNPINNED (no MATLAB here): follows
    MATLAB's documented defaults -- Gaussian sigma 1.5, dynamic range 1 for double images, K = (0.01, 0.03)."""
    A = np.asarray(recon_u8, dtype=np.float64) / 255.0
    R = np.asarray(gt_u8, dtype=np.float64) / 255.0
    m3 = _mask3(np.asarray(mask_u8, dtype=np.float64) / 255.0)
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    mux, muy = _gauss3d_replicate(A), _gauss3d_replicate(R)
    sx = _gauss3d_replicate(A * A) - mux * mux
    sy = _gauss3d_replicate(R * R) - muy * muy
    sxy = _gauss3d_replicate(A * R) - mux * muy
    ssim_map = ((2 * mux * muy + C1) * (2 * sxy + C2)) / ((mux * mux + muy * muy + C1) * (sx + sy + C2))
    return float((1.0 - (ssim_map * m3).sum() / m3.sum()) / 2.0)
