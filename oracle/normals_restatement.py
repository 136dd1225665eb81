"""TEST INFRASTRUCTURE.  Restatement of kornia==0.4.1 `kornia.geometry.depth.depth_to_normals`.

PARITY UNPINNED: kornia 0.4.1 is named by the reference (README.md:32; call sites T8:8, T8:353,
TLT:353, S1:326, S8:326, SLT:325) but its source is neither vendored under /root/reference nor
installed in this image, and there is no network.  The algorithm below restates kornia 0.4.1's
published implementation from memory:

  depth_to_3d:       X = (u - cx)/fx * d,  Y = (v - cy)/fy * d,  Z = d   (u = column, v = row)
  spatial_gradient:  mode='sobel', order=1, normalized=True -> 3x3 Sobel / 8, replicate padding,
                     applied independently to X, Y, Z;  channel 0 = d/du, channel 1 = d/dv
  depth_to_normals:  normalize(cross(dP/du, dP/dv), dim=1)  (F.normalize, eps 1e-12)

dtype follows torch promotion: the reference passes an f64 camera matrix (T8:571-577) and f32 depth,
so (u - cx)/fx is f64 and everything downstream is f64.
"""
import torch
import torch.nn.functional as F


def depth_to_3d(depth: torch.Tensor, camera_matrix: torch.Tensor) -> torch.Tensor:
    B, _, H, W = depth.shape
    v, u = torch.meshgrid(torch.arange(H, dtype=depth.dtype), torch.arange(W, dtype=depth.dtype), indexing="ij")
    K = camera_matrix[:, None, None]  # (B|1,1,1,3,3)
    fx, fy, cx, cy = K[..., 0, 0], K[..., 1, 1], K[..., 0, 2], K[..., 1, 2]
    x = (u[None] - cx) / fx
    y = (v[None] - cy) / fy
    xyz = torch.stack([x, y, torch.ones_like(x)], dim=-1)  # (B|1,H,W,3)
    pts = xyz * depth.permute(0, 2, 3, 1)
    return pts.permute(0, 3, 1, 2)


def spatial_gradient(x: torch.Tensor) -> torch.Tensor:
    B, C, H, W = x.shape
    kx = torch.tensor([[-1.0, 0.0, 1.0], [-2.0, 0.0, 2.0], [-1.0, 0.0, 1.0]], dtype=x.dtype) / 8.0
    ky = kx.t().contiguous()
    k = torch.stack([kx, ky])[:, None]  # (2,1,3,3)
    xp = F.pad(x.reshape(B * C, 1, H, W), [1, 1, 1, 1], mode="replicate")
    return F.conv2d(xp, k).view(B, C, 2, H, W)


def depth_to_normals(depth: torch.Tensor, camera_matrix: torch.Tensor) -> torch.Tensor:
    xyz = depth_to_3d(depth, camera_matrix)
    g = spatial_gradient(xyz)
    a, b = g[:, :, 0], g[:, :, 1]
    n = torch.cross(a, b, dim=1)
    return F.normalize(n, dim=1, p=2)
