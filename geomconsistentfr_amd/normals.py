"""Surface normals from depth: the product's restatement of kornia==0.4.1 `depth_to_normals`.

The reference calls kornia.geometry.depth.depth_to_normals(depth + 1610, K) and negates y
(train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:353-354).  kornia 0.4.1 is named in the reference's
README.md:32 but is neither vendored nor installed here, so this stage is PARITY-UNPINNED: it follows
kornia 0.4.1's published algorithm --
    P(u,v) = ((u-cx)/fx * d, (v-cy)/fy * d, d)                        depth_to_3d / unproject_points
    dP/du, dP/dv = normalised 3x3 Sobel (kernel/8), replicate padding   spatial_gradient(mode='sobel')
    n = normalize(cross(dP/du, dP/dv))                                   F.normalize, eps 1e-12
-- and everything downstream of it (shading given normals) is pinned by reference code.

`depth_to_normals` runs the HIP kernels gcfr_normals_fwd / gcfr_normals_bwd (csrc/gcfr_normals.hip) behind a
torch.autograd.Function.  It has NO CPU path: host tensors or a missing library raise GcfrError.  The torch-op
statement of the same algorithm the kernels are tested against lives with the test infrastructure
(oracle/normals_restatement.py).  The RelightNet mirrors do not call this op: they use block.render_from_depth,
where the same stencil is fused into the march kernel's epilogue and into the fused backward.
"""
import torch

from . import _lib


class _NormalsFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, depth, fx, fy, cx, cy, z_offset, negate_y):
        B, _, H, W = depth.shape
        d = depth.detach().to(torch.float32).contiguous()
        out = torch.empty((B, 3, H, W), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.load().gcfr_normals_fwd(d.data_ptr(), B, H, W, fx, fy, cx, cy, float(z_offset),
                                                    int(negate_y), out.data_ptr(),
                                                    torch.cuda.current_stream(d.device).cuda_stream), "gcfr_normals_fwd")
        ctx.save_for_backward(d)
        ctx.k = (fx, fy, cx, cy, float(z_offset), int(negate_y))
        return out

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        B, _, H, W = d.shape
        fx, fy, cx, cy, z_offset, negate_y = ctx.k
        g = g.detach().to(torch.float32).contiguous()
        gd = torch.zeros((B, 1, H, W), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _lib.check(_lib.load().gcfr_normals_bwd(g.data_ptr(), d.data_ptr(), B, H, W, fx, fy, cx, cy, z_offset,
                                                    negate_y, gd.data_ptr(),
                                                    torch.cuda.current_stream(d.device).cuda_stream), "gcfr_normals_bwd")
        return gd, None, None, None, None, None, None


def depth_to_normals(depth: torch.Tensor, camera_matrix: torch.Tensor, negate_y: bool = True,
                     z_offset: float = 0.0) -> torch.Tensor:
    """depth (B,1,H,W), camera_matrix (1|B,3,3) -> unit normals (B,3,H,W), y negated as the reference does
    right after the call (T8:354).  `z_offset` is added to the depth in f32 first (T8:353: +1610).
    Device tensors only (HIP kernels); there is no CPU path."""
    if not depth.is_cuda:
        raise _lib.GcfrError("geomconsistentfr_amd has no CPU path: depth must be on a ROCm device "
                             "(oracle/normals_restatement.py is the host-side specification used by the tests)")
    from .block import camera_scalars                            # host scalars, read once per tensor (no per-call sync)
    k4 = camera_scalars(camera_matrix)
    if k4 is None:
        return torch.cat([depth_to_normals(depth[i:i + 1], camera_matrix[i:i + 1], negate_y, z_offset)
                          for i in range(depth.shape[0])])
    fx, fy, cx, cy = k4
    return _NormalsFunction.apply(depth, fx, fy, cx, cy, z_offset, negate_y)
