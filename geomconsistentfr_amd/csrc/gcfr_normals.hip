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
    const float *z = a.depth + (size_t)b * P;
    const Grad3 g = point_gradients(a, z, r, c);
    const double cx_ = g.du[1] * g.dv[2] - g.du[2] * g.dv[1];
    const double cy_ = g.du[2] * g.dv[0] - g.du[0] * g.dv[2];
    const double cz_ = g.du[0] * g.dv[1] - g.du[1] * g.dv[0];
    const double nrm = sqrt(cx_ * cx_ + cy_ * cy_ + cz_ * cz_);
    const double nn = nrm > 1e-12 ? nrm : 1e-12;
    const double n0 = cx_ / nn, n1 = cy_ / nn, n2 = cz_ / nn;
    const float *gn = a.grad_normals + (size_t)b * 3 * P + p;
    const double g0 = gn[0], g1 = a.negate_y ? -(double)gn[P] : (double)gn[P], g2 = gn[2 * P];
    // n = c/|c|  (if |c| <= eps the denominator is the constant eps)
    double dc0, dc1, dc2;
    if (nrm > 1e-12) {
        const double ng = n0 * g0 + n1 * g1 + n2 * g2;
        dc0 = (g0 - n0 * ng) / nn;
        dc1 = (g1 - n1 * ng) / nn;
        dc2 = (g2 - n2 * ng) / nn;
    } else {
        dc0 = g0 / nn;
        dc1 = g1 / nn;
        dc2 = g2 / nn;
    }
    // c = du x dv:  d(du) = dv x dc,  d(dv) = dc x du
    const double ddu[3] = {g.dv[1] * dc2 - g.dv[2] * dc1, g.dv[2] * dc0 - g.dv[0] * dc2, g.dv[0] * dc1 - g.dv[1] * dc0};
    const double ddv[3] = {dc1 * g.du[2] - dc2 * g.du[1], dc2 * g.du[0] - dc0 * g.du[2], dc0 * g.du[1] - dc1 * g.du[0]};
    float *gz = a.grad_depth + (size_t)b * P;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const double ku = kSobelU[dr + 1][dc + 1], kv = kSobelV[dr + 1][dc + 1];
            if (ku == 0.0 && kv == 0.0)
                continue;
            const int rr = min(max(r + dr, 0), a.H - 1), cc = min(max(c + dc, 0), a.W - 1);
            const double ax = ((double)cc - a.cx) / a.fx, ay = ((double)rr - a.cy) / a.fy;
            // P_j = (ax*d, ay*d, d):  dd_j = ax*dP_x + ay*dP_y + dP_z,  dP = ku*d(du) + kv*d(dv)
            const double dPx = ku * ddu[0] + kv * ddv[0];
            const double dPy = ku * ddu[1] + kv * ddv[1];
            const double dPz = ku * ddu[2] + kv * ddv[2];
            atomicAdd(gz + (size_t)rr * a.W + cc, (float)(ax * dPx + ay * dPy + dPz));
        }
    }
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
    NormalsArgs a{depth, normals, grad_normals, grad_depth, H, W, fx, fy, cx, cy, z_offset, negate_y};
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
