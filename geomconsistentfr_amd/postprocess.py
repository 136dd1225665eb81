"""Callers and data formats either side of the render block -- device side and file formats.

Device path (HIP kernels of csrc/gcfr_postprocess.hip, no CPU fallback): `inference_images_device`,
`fix_border_artifacts_device` -- what the inference mirrors use, so a relit batch leaves the GPU as bytes:

  inference_images_device       S1:614-620 / S8:583-608 / SLT:547-579: composite + the five diagnostic maps, quantised
                                to the bytes cv2.imwrite stores
  fix_border_artifacts_device   fix_border_artifacts_CVPR2022.m:1-18

Host side: the on-disk formats of load_data() (T8:545-556).  The numpy statements of the script lines these kernels
implement live in oracle/postprocess_statements.py (test infrastructure, pinned to the reference's own main() by
tests/golden/slt_main_*.npz); nothing here imports them.
"""
import numpy as np


# ------------------------------------------------------------------------------------------------
# device path (csrc/gcfr_postprocess.hip)
# ------------------------------------------------------------------------------------------------
def inference_images_device(input_images, rendered, mask_u8, albedo=None, depth=None, shadow_mask_weights=None,
                            final_shading=None, surface_normals=None, mask_f32=False):
    """The images S1:614-620 / S8:603-608 / SLT:574-579 write, as uint8 device tensors (RGB, HWC), straight from the
    forward's device outputs: `rendered_image` always, the five diagnostic maps for whichever inputs are given.
    input_images (B,H,W,3) f32 in [0,1]; rendered / albedo / surface_normals (B,3,H,W); depth (B,1,H,W) or (B,H,W);
    shadow_mask_weights / final_shading (B,H,W); mask_u8 (1|B,H,W) or (H,W) uint8 skin mask as stored on disk (the kernel
    forms the scripts' mask/255.0 itself: f64 as S1:580 / S8:569 hold it, or with mask_f32=True f32 as SLT:540 does).
    Many lights per face: rendered (B,L,3,H,W) (and shadow_mask_weights / final_shading (B,L,H,W)) give `rendered_image`
    (B,L,H,W,3) (`shadow_mask` / `shading` (B,L,H,W)) from ONE launch that reads each photograph once per light in place --
    no L-fold copy of the input; the per-photograph maps (albedo, depth, normals) keep their (B,...) shapes."""
    import torch
    from . import _lib
    L_ = _lib.load()
    x = input_images
    if not x.is_cuda:
        raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: tensors must be on a ROCm device")
    f = lambda t: None if t is None else t.detach().to(torch.float32).contiguous()
    x, rendered = f(x), f(rendered)
    B, H, W, _ = x.shape
    multi = rendered.dim() == 5
    L = rendered.shape[1] if multi else 1
    lead = (B, L) if multi else (B,)
    if tuple(rendered.shape) != lead + (3, H, W):
        raise _lib.GcfrError("rendered must be (B,3,H,W) or (B,L,3,H,W) for input images %s; got %s" % (tuple(x.shape), tuple(rendered.shape)))
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
    out = {"rendered_image": u8(*lead, H, W, 3)}
    if shadow_mask_weights is not None:
        out["shadow_mask"] = u8(*lead, H, W)
    if albedo is not None:
        out["albedo"] = u8(B, H, W, 3)
    if depth is not None:
        out["depth"] = u8(B, H, W)
    if final_shading is not None:
        out["shading"] = u8(*lead, H, W)
    if surface_normals is not None:
        out["surface_normals"] = u8(B, H, W, 3)
    p = lambda t: None if t is None else t.data_ptr()
    with torch.cuda.device(x.device):
        _lib.check(L_.gcfr_inference_images_u8(
            x.data_ptr(), rendered.data_ptr(), p(albedo), p(depth), p(drange), p(shadow_mask_weights), p(final_shading),
            p(surface_normals), m.data_ptr(), m.shape[0], B, L, H, W, out["rendered_image"].data_ptr(), p(out.get("shadow_mask")),
            p(out.get("albedo")), p(out.get("depth")), p(out.get("shading")), p(out.get("surface_normals")), int(bool(mask_f32)),
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
