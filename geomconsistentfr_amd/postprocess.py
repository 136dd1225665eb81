"""Callers and data formats either side of the render block.

Device path (HIP kernels of csrc/gcfr_postprocess.hip, no CPU fallback): `inference_images_device`,
`fix_border_artifacts_device` -- what the inference mirrors use, so a relit batch leaves the GPU as bytes.
Host side, numpy (file formats, offline metrics, and the host statements of the two device functions that
the parity tests compare them with):

What the reference's inference scripts do with the block's outputs, and how its training script reads
its inputs, restated as functions (the scripts themselves -- argv parsing, hard-coded paths, PNG
writing loops -- are out of scope):

  composite_into_input     test_relight_single_image.py:614-620 (S1) / S8:596-602
  diagnostic_images        test_raytracing_relighting_CelebAHQ_DSSIM_8x.py:583-608 (S8): the six images per face
  to_uint8                 what cv2.imwrite does to a float image (saturating round-to-nearest-even)
  fix_border_artifacts     fix_border_artifacts_CVPR2022.m:1-18 (3x3 median on the 7x7-box mask border)
  masked_mse               MSE_MP.m:24
  masked_dssim             DSSIM_MP_RGB.m:24-26 (MATLAB ssim: PARITY UNPINNED, MATLAB is not available)
  load_depth_mat / load_lighting_mat / fill_nose_and_mouth_mask      load_data(), T8:545-556

Images are RGB, HWC, float in [0,1] unless a name says u8; the reference's BGR flips exist only because it
writes through cv2 and are not reproduced.
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
    """Paste the relit face into the input photograph (S1:614-620).
    input_image (H,W,3) in [0,1]; rendered (3,H,W) (one item of rendered_images); mask (H,W) in [0,1]
    (the reference divides the 4-level skin mask by 255).  Returns (H,W,3) float in [0,255]."""
    m3 = _mask3(mask)
    out = np.asarray(input_image, dtype=np.float64) * 255.0
    # 255.0*rendered_images[k] is an f32 product in the scripts (numpy keeps the array's dtype), widened by the f64 mask
    ren = (np.float32(255.0) * np.transpose(np.asarray(rendered, dtype=np.float32), (1, 2, 0))).astype(np.float64) * m3
    sel = m3 > 0
    out[sel] = ren[sel]
    return out


def diagnostic_images(input_image, albedo, depth_batch, index, shadow_mask_weights, rendered, final_shading,
                      surface_normals, mask) -> Dict[str, np.ndarray]:
    """The six images S8:603-608 writes per face (float, [0,255], RGB / single channel).
    depth_batch (B,1,H,W) is needed whole: the reference min-max normalises -depth over the BATCH (S8:589-590);
    the other arguments are item `index` of the forward's outputs in their native layouts:
    albedo (3,H,W), shadow_mask_weights (H,W), rendered (3,H,W), final_shading (H,W), surface_normals (3,H,W)."""
    m3 = _mask3(mask)
    m1 = m3[..., 0]
    f32, k255 = np.float32, np.float32(255.0)
    d = -np.asarray(depth_batch, dtype=f32)                       # the scripts' arrays are f32 until the f64 mask
    d = (d - d.min()) / (d.max() - d.min())
    hwc = lambda a: np.transpose(np.asarray(a, dtype=f32), (1, 2, 0))
    wide = lambda a: a.astype(np.float64)
    return {
        "rendered_image": composite_into_input(input_image, rendered, mask),
        "shadow_mask": wide(k255 * np.asarray(shadow_mask_weights, dtype=f32)) * m1,
        "albedo": wide(k255 * hwc(albedo)) * m3,
        "depth": wide(k255 * d[index, 0]) * m1,
        "shading": wide(k255 * np.asarray(final_shading, dtype=f32)) * m1,
        "surface_normals": wide(k255 * (hwc(surface_normals) + f32(1.0)) / f32(2.0)) * m3,
    }


# ------------------------------------------------------------------------------------------------
# device path (csrc/gcfr_postprocess.hip)
# ------------------------------------------------------------------------------------------------
def inference_images_device(input_images, rendered, mask_u8, albedo=None, depth=None, shadow_mask_weights=None,
                            final_shading=None, surface_normals=None):
    """The images S1:614-620 / S8:603-608 / SLT:574-579 write, as uint8 device tensors (RGB, HWC), straight from the
    forward's device outputs: `rendered_image` always, the five diagnostic maps for whichever inputs are given.
    input_images (B,H,W,3) f32 in [0,1]; rendered / albedo / surface_normals (B,3,H,W); depth (B,1,H,W) or (B,H,W);
    shadow_mask_weights / final_shading (B,H,W); mask_u8 (1|B,H,W) or (H,W) uint8 skin mask as stored on disk (the kernel
    forms the scripts' f64 mask/255.0 itself)."""
    import torch
    from . import _lib
    L_ = _lib.load()
    x = input_images
    if not x.is_cuda:
        raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: tensors must be on a ROCm device")
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    x, rendered = f(x), f(rendered)
    B, H, W, _ = x.shape
    if mask_u8.dtype != torch.uint8:
        raise _lib.GcfrError("mask_u8 must be the uint8 skin mask (0..255)")
    m = mask_u8.to(x.device).contiguous().reshape(-1, H, W)
    albedo, shadow_mask_weights, final_shading, surface_normals = f(albedo), f(shadow_mask_weights), f(final_shading), f(surface_normals)
    drange = None
    if depth is not None:
        depth = f(depth).reshape(B, H, W)
        neg = -depth
        drange = torch.stack([neg.amin(), neg.amax()]).contiguous()              # stays on the device: no sync
    u8 = lambda *shape: torch.empty(shape, dtype=torch.uint8, device=x.device)
    out = {"rendered_image": u8(B, H, W, 3)}
    if shadow_mask_weights is not None:
        out["shadow_mask"] = u8(B, H, W)
    if albedo is not None:
        out["albedo"] = u8(B, H, W, 3)
    if depth is not None:
        out["depth"] = u8(B, H, W)
    if final_shading is not None:
        out["shading"] = u8(B, H, W)
    if surface_normals is not None:
        out["surface_normals"] = u8(B, H, W, 3)
    p = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(L_.gcfr_inference_images_u8(
            x.data_ptr(), rendered.data_ptr(), p(albedo), p(depth), p(drange), p(shadow_mask_weights), p(final_shading),
            p(surface_normals), m.data_ptr(), m.shape[0], B, H, W, out["rendered_image"].data_ptr(), p(out.get("shadow_mask")),
            p(out.get("albedo")), p(out.get("depth")), p(out.get("shading")), p(out.get("surface_normals")),
            torch.cuda.current_stream(x.device).cuda_stream), "gcfr_inference_images_u8")
    return out


def fix_border_artifacts_device(img_u8, face_mask_u8):
    """fix_border_artifacts_CVPR2022.m on the device: img_u8 (B,H,W,3) uint8, face_mask_u8 (1|B,H,W) or (H,W) uint8
    skin mask as stored on disk.  Returns a new (B,H,W,3) uint8 tensor."""
    import torch
    from . import _lib
    if not img_u8.is_cuda:
        raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: tensors must be on a ROCm device")
    img = img_u8.contiguous()
    B, H, W, _ = img.shape
    mk = face_mask_u8.to(img.device, torch.uint8).contiguous().reshape(-1, H, W)
    out = torch.empty_like(img)
    with torch.cuda.device(img.device):
        _lib.check(_lib.load().gcfr_fix_border_u8(img.data_ptr(), mk.data_ptr(), mk.shape[0], B, H, W, out.data_ptr(),
                                                  torch.cuda.current_stream(img.device).cuda_stream), "gcfr_fix_border_u8")
    return out


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


# ------------------------------------------------------------------------------------------------
# on-disk formats of load_data() (T8:527-558)
# ------------------------------------------------------------------------------------------------
def load_depth_mat(path: str) -> np.ndarray:
    """`.mat` with key 'depth_img' -> (256,256,1) float64 (T8:545)."""
    import scipy.io
    d = scipy.io.loadmat(path)["depth_img"]
    return np.reshape(np.asarray(d, dtype=np.float64), (d.shape[0], d.shape[1], 1))


def load_lighting_mat(path: str, ambient: float = 0.5) -> np.ndarray:
    """`.mat` with key 'lighting_direction' -> [ambient, lx, ly, lz] (T8:542, 549: ambient target fixed at 0.5)."""
    import scipy.io
    l = np.asarray(scipy.io.loadmat(path)["lighting_direction"], dtype=np.float64).reshape(3)
    return np.concatenate([[ambient], l])


def fill_nose_and_mouth_mask(face_mask_u8: np.ndarray, depth_mask_u8: np.ndarray) -> np.ndarray:
    """T8:552-556: element-wise max of the two masks, then > 128 -> 255, else 0.  Returns float64 (H,W,1)."""
    t = np.maximum(np.asarray(face_mask_u8, dtype=np.float64), np.asarray(depth_mask_u8, dtype=np.float64))
    t = np.where(t > 128, 255.0, 0.0)
    return t.reshape(t.shape[0], t.shape[1], 1)
