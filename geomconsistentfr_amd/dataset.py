"""The training script's dataset, kept as bytes.

`load_data()` of the reference (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:527-558) reads six directories into
float64 numpy arrays -- 29,890 faces x (256x256x3 + 4 + 256x256 x 4) doubles = 110 GB of host memory -- and the loop
slices batches of them, dividing by 255 again where it did not do so when loading (T8:607-615).  Here the same files
are read into uint8 arrays (7.3 GB for the full set; depth maps f32, +7.8 GB), a batch is uploaded as bytes and
converted by one HIP kernel (`gcfr_assemble_batch_u8`, csrc/gcfr_dataset.hip) with the script's arithmetic; the result
is the dict `Trainer.step` consumes.

Directory layout and pairing rules are the script's own:
  depth_maps_CelebA-HQ/<id>_*.mat                      key 'depth_img'                                       T8:539, 545
  depth_masks_CelebA-HQ_DFNRMVS/*                      paired with the depth maps BY POSITION in the sorted lists   T8:540, 546
  lighting_directions_CelebAHQ_DFNRMVS/<id>.jpg.mat    key 'lighting_direction'; <id> = depth file name up to '_'   T8:548-549
  CelebA-HQ_DFNRMVS_cropped/<id>.jpg                   image                                                 T8:550
  CelebA-HQ_albedo_grayscale/<id>.jpg                  grey albedo                                           T8:551
  CelebAHQ_face_masks/<id>.jpg                         face mask, for the fill-nose-and-mouth mask           T8:552-556
The ambient target is the constant 0.5 (T8:542).
"""
import os
from typing import Dict, Optional, Sequence

import numpy as np
import torch

from . import _lib

DIRS = dict(images="CelebA-HQ_DFNRMVS_cropped", lightings="lighting_directions_CelebAHQ_DFNRMVS", depths="depth_maps_CelebA-HQ",
            masks="depth_masks_CelebA-HQ_DFNRMVS", albedo="CelebA-HQ_albedo_grayscale", face_masks="CelebAHQ_face_masks")


def _imread(path: str) -> np.ndarray:
    """imageio.imread for the formats the script reads (jpg / png): Pillow's decoder, as imageio uses for both."""
    from PIL import Image
    with Image.open(path) as im:
        return np.asarray(im)


class RelightDataset:
    """The six arrays of load_data(), uint8 where the files are uint8.  `root` is the script's 'MP_data/' directory."""

    def __init__(self, root: str, limit: Optional[int] = None, H: int = 256, W: int = 256):
        import scipy.io
        d = {k: os.path.join(root, v) for k, v in DIRS.items()}
        depths = sorted(os.listdir(d["depths"]))                                           # T8:539
        masks = sorted(os.listdir(d["masks"]))                                             # T8:540
        if len(masks) < len(depths):       # the script iterates range(len(depths)) and indexes masks[i]: surplus masks are tolerated
            raise ValueError("load_data pairs depth maps and depth masks by position: %d maps, only %d masks" % (len(depths), len(masks)))
        n = len(depths) if limit is None else min(limit, len(depths))
        self.H, self.W, self.ids = H, W, []
        self.images = np.zeros((n, H, W, 3), np.uint8)
        self.lightings = np.zeros((n, 4), np.float32)
        self.depths = np.zeros((n, H, W, 1), np.float32)
        self.masks = np.zeros((n, H, W), np.uint8)
        self.albedo = np.zeros((n, H, W), np.uint8)
        self.face_masks = np.zeros((n, H, W), np.uint8)
        self.lightings[:, 0] = 0.5                                                         # T8:542
        for i in range(n):
            ident = depths[i].split("_")[0]                                                # T8:548
            self.ids.append(ident)
            self.depths[i] = np.reshape(scipy.io.loadmat(os.path.join(d["depths"], depths[i]))["depth_img"], (H, W, 1))
            self.masks[i] = np.reshape(_imread(os.path.join(d["masks"], masks[i])), (H, W))
            self.lightings[i, 1:4] = np.reshape(scipy.io.loadmat(os.path.join(d["lightings"], ident + ".jpg.mat"))["lighting_direction"], 3)
            self.images[i] = _imread(os.path.join(d["images"], ident + ".jpg"))
            self.albedo[i] = _imread(os.path.join(d["albedo"], ident + ".jpg"))
            self.face_masks[i] = np.reshape(_imread(os.path.join(d["face_masks"], ident + ".jpg")), (H, W))

    def __len__(self) -> int:
        return len(self.ids)

    def host_bytes(self) -> int:
        return sum(a.nbytes for a in (self.images, self.lightings, self.depths, self.masks, self.albedo, self.face_masks))

    def batch(self, indices: Sequence[int], device="cuda") -> Dict[str, torch.Tensor]:
        """The batch dict of `Trainer.step` for faces `indices` (T8:607-615): bytes go up, floats are made on the device."""
        idx = np.asarray(indices, dtype=np.int64)
        dev = torch.device(device)
        if dev.type != "cuda":
            raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: RelightDataset.batch needs a ROCm device")
        up = lambda a: torch.from_numpy(np.ascontiguousarray(a[idx])).to(dev, non_blocking=True)
        out = assemble_batch(up(self.images), up(self.masks), up(self.face_masks), up(self.albedo))
        out["lightings"] = up(self.lightings)
        out["depths"] = up(self.depths)
        return out


def assemble_batch(images_u8: torch.Tensor, depth_mask_u8: torch.Tensor, face_mask_u8: torch.Tensor,
                   albedo_u8: torch.Tensor) -> Dict[str, torch.Tensor]:
    """uint8 device tensors (B,H,W,3), (B,H,W) x 3 -> {images (B,H,W,3), masks, masks_fill, albedo (B,H,W,1)} f32, with the
    arithmetic of T8:550-556 and :610-615 (`gcfr_assemble_batch_u8`)."""
    for t in (images_u8, depth_mask_u8, face_mask_u8, albedo_u8):
        if not t.is_cuda or t.dtype != torch.uint8:
            raise _lib.GcfrError("assemble_batch: uint8 tensors on a ROCm device (there is no CPU path)")
    L_ = _lib.load()
    x, dm, fm, al = (t.contiguous() for t in (images_u8, depth_mask_u8, face_mask_u8, albedo_u8))
    B, H, W, _ = x.shape
    f32 = dict(dtype=torch.float32, device=x.device)
    images = torch.empty((B, H, W, 3), **f32)
    masks, fill, albedo = (torch.empty((B, H, W, 1), **f32) for _ in range(3))
    with torch.cuda.device(x.device):
        _lib.check(L_.gcfr_assemble_batch_u8(x.data_ptr(), dm.data_ptr(), fm.data_ptr(), al.data_ptr(), B, H, W, images.data_ptr(),
                                             masks.data_ptr(), fill.data_ptr(), albedo.data_ptr(),
                                             torch.cuda.current_stream(x.device).cuda_stream), "gcfr_assemble_batch_u8")
    return dict(images=images, masks=masks, masks_fill=fill, albedo=albedo)


def masked_metrics(recon_u8: torch.Tensor, gt_u8: torch.Tensor, mask_u8: torch.Tensor):
    """MSE_MP.m:24 and DSSIM_MP_RGB.m:24-26 for B image pairs on the device: recon_u8, gt_u8 (B,H,W,3) uint8 RGB,
    mask_u8 (B|1,H,W) or (H,W) uint8.  Returns (mse (B,), dssim (B,)) float64 device tensors (`gcfr_masked_metrics_u8`;
    DSSIM follows MATLAB ssim()'s documented defaults -- unpinned, see include/gcfr.h)."""
    for t in (recon_u8, gt_u8, mask_u8):
        if not t.is_cuda or t.dtype != torch.uint8:
            raise _lib.GcfrError("masked_metrics: uint8 tensors on a ROCm device (there is no CPU path)")
    L_ = _lib.load()
    r, g = recon_u8.contiguous(), gt_u8.contiguous()
    B, H, W, _ = r.shape
    m = mask_u8.contiguous().reshape(-1, H, W)
    ws_bytes = int(L_.gcfr_masked_metrics_workspace_bytes(B, H, W))
    ws = torch.empty(ws_bytes // 8, dtype=torch.float64, device=r.device)
    mse = torch.empty(B, dtype=torch.float64, device=r.device)
    dssim = torch.empty(B, dtype=torch.float64, device=r.device)
    with torch.cuda.device(r.device):
        _lib.check(L_.gcfr_masked_metrics_u8(r.data_ptr(), g.data_ptr(), m.data_ptr(), m.shape[0], B, H, W, mse.data_ptr(),
                                             dssim.data_ptr(), ws.data_ptr(), ws_bytes,
                                             torch.cuda.current_stream(r.device).cuda_stream), "gcfr_masked_metrics_u8")
    return mse, dssim
