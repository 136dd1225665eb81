"""Surface normals from depth: the product's restatement of kornia==0.4.1 `depth_to_normals`.

The reference calls kornia.geometry.depth.depth_to_normals(depth + 1610, K) and negates y
(train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:353-354).  kornia 0.4.1 is named in the reference's
README.md:32 but is neither vendored nor installed here, so this stage is PARITY-UNPINNED: it follows
kornia 0.4.1's published algorithm --
    P(u,v) = ((u-cx)/fx * d, (v-cy)/fy * d, d)                        depth_to_3d / unproject_points
    dP/du, dP/dv = normalised 3x3 Sobel (kernel/8), replicate padding   spatial_gradient(mode='sobel')
    n = normalize(cross(dP/du, dP/dv))                                   F.normalize, eps 1e-12
-- and everything downstream of it (shading given normals) is pinned by reference code.

Plain torch ops (device-agnostic, differentiable); SURVEY.md 8(f)-1 lists fusing this stencil into the
HIP shading kernel as the next widening step.
"""
import torch
import torch.nn.functional as F


def _sobel(p):
    """p (B,C,H,W) -> (d/du, d/dv), each (B,C,H,W); normalised Sobel with replicate padding."""
    q = F.pad(p, [1, 1, 1, 1], mode="replicate")
    tl, tc, tr = q[..., :-2, :-2], q[..., :-2, 1:-1], q[..., :-2, 2:]
    ml, mr = q[..., 1:-1, :-2], q[..., 1:-1, 2:]
    bl, bc, br = q[..., 2:, :-2], q[..., 2:, 1:-1], q[..., 2:, 2:]
    du = ((tr - tl) + 2.0 * (mr - ml) + (br - bl)) / 8.0
    dv = ((bl - tl) + 2.0 * (bc - tc) + (br - tr)) / 8.0
    return du, dv


def depth_to_normals(depth: torch.Tensor, camera_matrix: torch.Tensor, negate_y: bool = True) -> torch.Tensor:
    """depth (B,1,H,W), camera_matrix (1|B,3,3) -> unit normals (B,3,H,W) in depth's dtype, y negated
    as the reference does right after the call (T8:354)."""
    B, _, H, W = depth.shape
    K = camera_matrix.to(device=depth.device)
    ct = torch.promote_types(depth.dtype, K.dtype)            # the reference's K is f64 -> f64 maths
    d = depth.to(ct)
    u = torch.arange(W, dtype=ct, device=depth.device).view(1, 1, 1, W)
    v = torch.arange(H, dtype=ct, device=depth.device).view(1, 1, H, 1)
    fx, fy = K[:, 0, 0].view(-1, 1, 1, 1).to(ct), K[:, 1, 1].view(-1, 1, 1, 1).to(ct)
    cx, cy = K[:, 0, 2].view(-1, 1, 1, 1).to(ct), K[:, 1, 2].view(-1, 1, 1, 1).to(ct)
    pts = torch.cat([(u - cx) / fx * d, (v - cy) / fy * d, d], dim=1)
    du, dv = _sobel(pts)
    n = F.normalize(torch.cross(du, dv, dim=1), dim=1, p=2)
    if negate_y:
        n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], dim=1)
    return n.to(depth.dtype)
