"""TEST INFRASTRUCTURE -- "materialised-tensor" torch-CPU port of the reference render block.

Why a second oracle next to the C one (oracle/gcfr_oracle.c):
  * autograd through this port yields the reference's gradients (same op graph), which is what the
    HIP backward kernels are checked against at sizes the committed golden grads do not cover;
  * it has the reference's *performance character* (every (N,2,H,W) intermediate is materialised,
    images are processed one after another), so it is what bench.py times as `cpu_baseline`
    (kind "port") -- the reference's own .py cannot travel to the GPU box.

It follows train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:352-524 ("T8") op for op -- same dtypes,
same promotion points (f64 sample table), same separately-rounded mul/add -- but is generalised over
batch, H, W, N and the inference variants of SURVEY.md Appendix B, and is written as straight-line
tensor code instead of the reference's nine-way Python branch.  Pinned against the reference itself
in tests/test_oracle_vs_reference.py (authoring container) and against tests/golden/ everywhere.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class BlockParams:
    """Constants of the render block.  Defaults = T8."""
    n_samples: int = 160               # T8:48
    t0: float = 0.025                  # T8:468
    dt: float = 0.005
    light_distance: float = 4013.0     # T8:47
    directional_intensity: float = 0.5  # T8:46
    clamp_light_z_min: Optional[float] = 0.0   # T8:358; None = target light, no clamp (S1:332)
    inside_bonus: float = 0.0          # S1:495-496 -> 5.0
    bonus_box: Optional[Tuple[float, float, float, float]] = None  # (x_lo,x_hi,y_lo,y_hi)

    def sample_table(self) -> np.ndarray:
        delta = (self.t0 + self.dt) - self.t0       # np.arange value rule
        return self.t0 + np.arange(self.n_samples, dtype=np.float64) * delta


def pixel_grids(H: int, W: int):
    """xx = col - W/2, yy = H/2 - row (T8:51-55), shape (H,W) f32."""
    cols = torch.arange(W, dtype=torch.float32)
    rows = torch.arange(H, dtype=torch.float32)
    xx = (cols - W / 2.0)[None, :].expand(H, W)
    yy = (H / 2.0 - rows)[:, None].expand(H, W)
    return xx.contiguous(), yy.contiguous()


def light_points(light: torch.Tensor, p: BlockParams):
    """(B,3) raw/target light -> unit direction (B,3), light point C (B,3).  T8:357-362 / S1:332-335."""
    if p.clamp_light_z_min is not None:
        lz = torch.maximum(light[:, 2], torch.tensor(p.clamp_light_z_min, dtype=light.dtype))
        light = torch.stack([light[:, 0], light[:, 1], lz], dim=1)
    unit = F.normalize(light, p=2, dim=1)
    return unit, p.light_distance * unit


def _end_points(x, y, C, H, W):
    """Segment end point on the image box for one image.  T8:378-465.  x,y (H,W) f32; C (3,) f32."""
    x_lo, x_hi = -(W / 2.0), (W - W / 2.0 - 1)
    y_lo, y_hi = 1 - (H / 2.0), H / 2.0
    m = (C[1] - y) / (C[0] - x + 0.0001)                      # T8:378
    ic = C[1] - m * C[0]                                      # T8:379
    LX, LY = float(C[0].detach()), float(C[1].detach())       # T8:380-381 (python floats: no grad)
    one = torch.ones_like(x)

    def on_x(xb):                                             # "try x = xb": (xb, m*xb + ic)
        xs = xb * one
        return torch.stack([xs, m * xs + ic])

    def on_y(yb):                                             # "try y = yb": ((yb - ic)/(m + 1e-4), yb)
        ys = yb * one
        xs = (ys - ic) / (m + 0.0001)
        return torch.stack([xs, ys]), xs

    xin = x_lo <= LX <= x_hi
    yin = y_lo <= LY <= y_hi
    if xin and yin:                                           # T8:422-425
        E = torch.stack([LX * one, LY * one])
    elif xin:                                                 # T8:417-421, 426-430
        E, _ = on_y(y_lo if LY < y_lo else y_hi)
    elif yin:                                                 # T8:399-403, 444-448
        E = on_x(x_lo if LX < x_lo else x_hi)
    else:                                                     # corner cases T8:387-398 etc.
        EX = on_x(x_lo if LX < x_lo else x_hi)
        EY, xs = on_y(y_lo if LY < y_lo else y_hi)
        hit = torch.logical_and(xs >= x_lo, xs <= x_hi)
        E = EY * hit + EX * torch.logical_not(hit)            # arithmetic select T8:398
    # clamp T8:462-465 (masked assignment: zero gradient where clamped)
    Ex = torch.where(E[0] < x_lo, torch.full_like(E[0], x_lo), E[0])
    Ex = torch.where(Ex > x_hi, torch.full_like(Ex, x_hi), Ex)
    Ey = torch.where(E[1] < y_lo, torch.full_like(E[1], y_lo), E[1])
    Ey = torch.where(Ey > y_hi, torch.full_like(Ey, y_hi), Ey)
    return torch.stack([Ex, Ey])


def min_distance_one(depth_hw, mask_hw, C, p: BlockParams, return_all: bool = False):
    """Minimum point-to-line distance over the sample table for ONE image.  T8:375-515.
    depth_hw (H,W) f32; mask_hw (H,W) any dtype (0 = outside); C (3,) f32.  -> (values (H,W), idx (H,W));
    return_all: also the (N,H,W) distances the minimum was taken over (T8:512), before any bonus."""
    H, W = depth_hw.shape
    N = p.n_samples
    xx, yy = pixel_grids(H, W)
    start = torch.stack([xx, yy])                                            # (2,H,W)
    E = _end_points(xx, yy, C, H, W)
    diff = E - start                                                         # T8:467
    t = torch.from_numpy(p.sample_table()).reshape(N, 1, 1, 1)               # f64, T8:468
    pos = start[None] + t * diff[None]                                       # (N,2,H,W) f64, T8:472/480
    # rounded cell (mask lookup) T8:472-477
    rc = torch.round(pos)
    col_r = (rc[:, 0] + W / 2.0).int().long().reshape(-1)
    row_r = (H / 2.0 - rc[:, 1]).int().long().reshape(-1)
    # unrounded T8:480-487
    ux = (pos[:, 0] + W / 2.0) - 0.0001
    uy = (H / 2.0 - pos[:, 1]) - 0.0001
    ux, uy = ux.reshape(-1), uy.reshape(-1)
    fx, gx = torch.floor(ux).int(), torch.ceil(ux).int()
    fy, gy = torch.floor(uy).int(), torch.ceil(uy).int()
    fxl, gxl, fyl, gyl = fx.long(), gx.long(), fy.long(), gy.long()
    zUL, zUR = depth_hw[fyl, fxl], depth_hw[fyl, gxl]                        # T8:488-491 (index -1 wraps)
    zLL, zLR = depth_hw[gyl, fxl], depth_hw[gyl, gxl]
    up = zUL * (gx - ux) + zUR * (ux - fx)                                   # T8:492
    low = zLL * (gx - ux) + zLR * (ux - fx)                                  # T8:493
    zA = up * (gy - uy) + low * (uy - fy)                                    # T8:494
    A = torch.stack([ux - W / 2.0, H / 2.0 - uy, zA]).reshape(3, N, H, W).float()   # T8:497-502
    Bp = torch.stack([xx, yy, depth_hw]).reshape(3, 1, H, W)
    BA = A - Bp                                                              # T8:504
    BC = (C.reshape(3, 1, 1, 1) - Bp).expand(3, N, H, W)                     # T8:505-507
    X = torch.cross(BA, BC, dim=0)                                           # T8:508
    d = torch.sqrt(torch.sum(X * X, dim=0) + 0.0001) / torch.sqrt(torch.sum(BC * BC, dim=0) + 0.0001)
    out = (mask_hw[row_r, col_r] == 0).reshape(N, H, W)                      # T8:510
    d = torch.logical_not(out) * d + out * 1000000.0                         # T8:512
    values, idx = torch.min(d, dim=0)                                        # T8:514
    if return_all:
        return values, idx, d
    if p.inside_bonus != 0.0 and p.bonus_box is not None:                    # S1:495-496
        bx0, bx1, by0, by1 = p.bonus_box
        LX, LY = float(C[0].detach()), float(C[1].detach())
        if bx0 <= LX <= bx1 and by0 <= LY <= by1:
            values = values + p.inside_bonus
    return values, idx


def shadow_transfer(min_dist):
    """T8:517: w = 1 - 4 e^-d / (1 + e^-d)^2."""
    return -4 * torch.exp(-min_dist) / torch.pow((1 + torch.exp(-min_dist)), 2) + 1


def render_block(depth, albedo, light, ambient, normals, masks, p: BlockParams = BlockParams()):
    """Whole block for a batch.
      depth (B,1,H,W) f32; albedo (B,3,H,W) f32; light (B,3) raw (T8) or target (S1); ambient (B,) f32;
      normals (B,3,H,W): depth_to_normals(depth+offset, K) with y negated (T8:353-354), not yet re-normalised;
      masks (B,H,W) or (1,H,W), 0 = outside.
    Returns dict with the reference's tensors (names as T8:524 / S1:505)."""
    B, _, H, W = depth.shape
    xx, yy = pixel_grids(H, W)
    pts = torch.cat([xx.expand(B, 1, H, W), yy.expand(B, 1, H, W), depth], 1)           # T8:356
    unit, C = light_points(light, p)
    Cmap = C.reshape(B, 3, 1, 1).expand(B, 3, H, W)
    inc = F.normalize(Cmap - pts, p=2, dim=1)                                           # T8:364
    nrm = F.normalize(normals, p=2, dim=1)                                              # T8:365
    directional = p.directional_intensity * torch.maximum(torch.sum(nrm * inc, dim=1), torch.tensor([0.0]))
    ambient_light = ambient.reshape(B, 1, 1).expand(B, H, W)                            # T8:367-368
    full = ambient_light + directional                                                  # T8:369
    md = []
    for i in range(B):                                                                  # T8:374
        mk = masks[i if masks.shape[0] == B else 0]
        v, _ = min_distance_one(depth[i, 0], mk, C[i], p)
        md.append(v)
    md = torch.stack(md)
    w = shadow_transfer(md)                                                             # T8:517
    final = w * full + (1 - w) * ambient_light                                          # T8:518
    rendered = (albedo * final[:, None]).to(albedo.dtype)                               # T8:519-522
    return dict(shadow_mask_weights=w, ambient_light=ambient_light, full_shading=full,
                rendered_images=rendered, unit_light_direction=unit.reshape(B, 3, 1, 1),
                ambient_values=ambient.reshape(B, 1, 1), final_shading=final, surface_normals=nrm,
                minimum_distance=md)
