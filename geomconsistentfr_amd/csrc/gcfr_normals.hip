// Surface normals from depth (kornia 0.4.1 `depth_to_normals` restatement), forward and backward, gfx950.
//
// Replaces the library call at train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:353 plus the sign flip
// at :354.  PARITY UNPINNED: kornia is named by the reference (README.md:32) but not vendored; the
// algorithm restated here is kornia 0.4.1's published one --
//     P(u,v) = ((u-cx)/fx * d, (v-cy)/fy * d, d)                       depth_to_3d
//     dP/du, dP/dv: 3x3 Sobel / 8 with replicate padding               spatial_gradient('sobel', normalized)
//     n = normalize(cross(dP/du, dP/dv)), eps 1e-12                    F.normalize
// -- see geomconsistentfr_amd/normals.py for the same statement in torch ops (the two are tested against
// each other and against oracle/normals_restatement.py).
//
// One lane per pixel; the 3x3 neighbourhood comes through L1/L2 (each depth value is read by nine
// lanes of neighbouring pixels); arithmetic in f64 as in the reference (its camera matrix is f64, so
// torch promotes the whole stage), output f32.  Streaming kernel: 4 B read + 12 B written per pixel.
#include "gcfr_device.hpp"

#include "../../include/gcfr.h"

namespace gcfr {

__global__ __launch_bounds__(256) void normals_fwd_kernel(NormalsArgs a)
{
    const size_t P = (size_t)a.H * a.W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P)
        return;
    const int r = (int)(p / a.W), c = (int)(p - (size_t)r * a.W);
    float n[3];
    unit_normal(a, a.depth + (size_t)b * P, r, c, n);
    float *o = a.normals + (size_t)b * 3 * P + p;
    o[0] = n[0];
    o[P] = n[1];
    o[2 * P] = n[2];
}

__global__ __launch_bounds__(256) void normals_bwd_kernel(NormalsArgs a)
{
    const size_t P = (size_t)a.H * a.W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P)
        return;
    const int r = (int)(p / a.W), c = (int)(p - (size_t)r * a.W);
    const float *gn = a.grad_normals + (size_t)b * 3 * P + p;
    normals_bwd_pixel(a, a.depth + (size_t)b * P, a.grad_depth + (size_t)b * P, r, c, gn[0], gn[P], gn[2 * P]);
}

}  // namespace gcfr

using namespace gcfr;

static int normals_launch(bool backward, const float *depth, const float *grad_normals, int32_t B, int32_t H,
                          int32_t W, double fx, double fy, double cx, double cy, float z_offset,
                          int32_t negate_y, float *normals, float *grad_depth, void *stream)
{
    if (!depth || B <= 0 || H <= 0 || W <= 0 || B > 65535 || fx == 0.0 || fy == 0.0)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (backward ? (!grad_normals || !grad_depth) : !normals)
        return GCFR_ERR_INVALID_ARGUMENT;
    NormalsArgs a{depth, normals, grad_normals, grad_depth, H, W, fx, fy, cx, cy, z_offset, negate_y, 0.0, 0.0};
    set_focal(a, fx, fy);
    const size_t P = (size_t)H * W;
    const dim3 grid((unsigned)((P + 255) / 256), (unsigned)B);
    if (backward)
        hipLaunchKernelGGL(normals_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(normals_fwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH;
}

extern "C" int gcfr_normals_fwd(const float *depth, int32_t B, int32_t H, int32_t W, double fx, double fy,
                                double cx, double cy, float z_offset, int32_t negate_y, float *normals,
                                void *stream)
{
    return normals_launch(false, depth, nullptr, B, H, W, fx, fy, cx, cy, z_offset, negate_y, normals, nullptr,
                          stream);
}

extern "C" int gcfr_normals_bwd(const float *grad_normals, const float *depth, int32_t B, int32_t H,
                                int32_t W, double fx, double fy, double cx, double cy, float z_offset,
                                int32_t negate_y, float *grad_depth, void *stream)
{
    return normals_launch(true, depth, grad_normals, B, H, W, fx, fy, cx, cy, z_offset, negate_y, nullptr,
                          grad_depth, stream);
}
