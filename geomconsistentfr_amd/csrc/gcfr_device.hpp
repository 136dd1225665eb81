// Device-side helpers shared by the gfx950 kernels of the GeomConsistentFR render block.
//
// Arithmetic contract: the reference is a chain of separately rounded torch
// elementwise ops, so this translation unit is compiled with -ffp-contract=off and uses an explicit
// fma only where the reference's own kernels do (torch.cross, vector 2-norm).  Decisions that pick
// array cells (round / floor / ceil of f64 sample positions) must match the reference exactly;
// values must match to <=1e-4 (shadow weight) and <=1e-3 (RGB).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace gcfr {

constexpr float kEps4 = 0.0001f;        // the reference's 0.0001 guards (T8:378, 395, 483, 509)
constexpr float kMaskedDistance = 1000000.0f;  // T8:512

// gfx9 raw buffer descriptor word 3 (DATA_FORMAT=32, no swizzle): out-of-range offsets read 0 and
// drop writes, so a corrupted index can never fault the GPU.
constexpr int kBufferRsrcWord3 = 0x00020000;

__device__ inline __amdgpu_buffer_rsrc_t make_rsrc(const void *p, int bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, kBufferRsrcWord3);
}
__device__ inline float buf_load_f32(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, byte_off, 0, 0));
}
__device__ inline uint32_t buf_load_u8(__amdgpu_buffer_rsrc_t r, int byte_off)
{
    return __builtin_amdgcn_raw_buffer_load_b8(r, byte_off, 0, 0);
}

// Image box in image-plane coordinates (T8:386-387, 416, 399).
struct Box {
    float x_lo, x_hi, y_lo, y_hi;
};
__device__ __host__ inline Box image_box(int H, int W)
{
    return Box{-(W / 2.0f), W - W / 2.0f - 1.0f, 1.0f - H / 2.0f, H / 2.0f};
}

// Which of the nine end-point branches a light selects (uniform per (image, light)).  T8:386-460.
struct LightCase {
    int xcase, ycase;  // 0: below lo, 1: inside [lo, hi], 2: above hi
};
__device__ inline LightCase classify_light(float Cx, float Cy, const Box &bx)
{
    LightCase lc;
    lc.xcase = (Cx < bx.x_lo) ? 0 : (Cx <= bx.x_hi ? 1 : 2);
    lc.ycase = (Cy < bx.y_lo) ? 0 : (Cy <= bx.y_hi ? 1 : 2);
    return lc;
}

// End point of the 2-D segment pixel -> light, clipped to the image box.  T8:378-465.
// All f32, each operation separately rounded, arithmetic (not boolean) selection at T8:398.
__device__ inline void end_point(float x, float y, float Cx, float Cy, const Box &bx, LightCase lc,
                                 float &Ex, float &Ey)
{
    const float m = (Cy - y) / ((Cx - x) + kEps4);  // slopes      T8:378
    const float ic = Cy - m * Cx;                   // intercepts  T8:379
    float ex, ey;
    if (lc.xcase == 1) {
        if (lc.ycase == 1) {  // light projects inside the image: its own xy (T8:422-425)
            ex = Cx;
            ey = Cy;
        } else {  // T8:417-421 / 426-430
            const float yb = (lc.ycase == 0) ? bx.y_lo : bx.y_hi;
            ex = (yb - ic) / (m + kEps4);
            ey = yb;
        }
    } else {
        const float xb = (lc.xcase == 0) ? bx.x_lo : bx.x_hi;
        const float Xy = m * xb + ic;  // T8:390
        if (lc.ycase == 1) {           // T8:399-403 / 444-448
            ex = xb;
            ey = Xy;
        } else {  // corner branches T8:387-398, 404-415, 432-443, 449-460
            const float yb = (lc.ycase == 0) ? bx.y_lo : bx.y_hi;
            const float Yx = (yb - ic) / (m + kEps4);
            const float b = (Yx >= bx.x_lo && Yx <= bx.x_hi) ? 1.0f : 0.0f;
            const float nb = 1.0f - b;
            ex = Yx * b + xb * nb;  // a non-finite candidate poisons the result, as in the reference
            ey = yb * b + Xy * nb;
        }
    }
    // clamp T8:462-465 (NaN passes through, as a masked assignment would leave it)
    ex = (ex < bx.x_lo) ? bx.x_lo : ex;
    ex = (ex > bx.x_hi) ? bx.x_hi : ex;
    ey = (ey < bx.y_lo) ? bx.y_lo : ey;
    ey = (ey > bx.y_hi) ? bx.y_hi : ey;
    Ex = ex;
    Ey = ey;
}

// torch's vector 2-norm accumulates acc = fma(v, v, acc) (probed; see oracle/gcfr_oracle.c).
__device__ inline float norm3_torch(float a, float b, float c)
{
    return __builtin_sqrtf(__builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a)));
}

// T8:517: w = 1 - 4 e^-d / (1 + e^-d)^2   (== tanh^2(d/2)); evaluated as written, in f32.
__device__ inline float shadow_transfer(float d)
{
    const float e = expf(-d);  // precise expf (the fast __expf is deliberately not used)
    const float onepe = 1.0f + e;
    return (-4.0f * e) / (onepe * onepe) + 1.0f;
}

// Shading of one pixel for one light, T8:364-369 and 517-518 (shared by the stand-alone shade kernel
// and the fused epilogue of the march kernel so that both produce the same bits).
struct Shaded {
    float w, full, fin;
};
__device__ inline Shaded shade_pixel(float x, float y, float zb, float nx, float ny, float nz, float Cx,
                                     float Cy, float Cz, float amb, float intensity, float min_dist)
{
    // incident light direction, T8:364
    const float lx = Cx - x, ly = Cy - y, lz = Cz - zb;
    float ln = norm3_torch(lx, ly, lz);
    ln = ln > 1e-12f ? ln : 1e-12f;
    const float ux = lx / ln, uy = ly / ln, uz = lz / ln;
    // surface normal, re-normalised (T8:365)
    float nn = norm3_torch(nx, ny, nz);
    nn = nn > 1e-12f ? nn : 1e-12f;
    const float n0 = nx / nn, n1 = ny / nn, n2 = nz / nn;
    const float dot = (n0 * ux + n1 * uy) + n2 * uz;  // T8:366
    Shaded o;
    o.full = amb + intensity * (dot > 0.0f ? dot : 0.0f);  // T8:366-369
    o.w = shadow_transfer(min_dist);                        // T8:517
    o.fin = o.w * o.full + (1.0f - o.w) * amb;              // T8:518
    return o;
}

// ---- surface normals from depth (kornia 0.4.1 restatement; see gcfr_normals.hip for the algorithm) ----
struct NormalsArgs {
    const float *depth;  // (B,H,W)
    float *normals;      // (B,3,H,W)           forward output
    const float *grad_normals;  // (B,3,H,W)    backward input
    float *grad_depth;   // (B,H,W) +=          backward output
    int32_t H, W;
    double fx, fy, cx, cy;
    float z_offset;
    int32_t negate_y;
};

// Sobel weights (already / 8) indexed [dr+1][dc+1]
__device__ constexpr double kSobelU[3][3] = {{-0.125, 0.0, 0.125}, {-0.25, 0.0, 0.25}, {-0.125, 0.0, 0.125}};
__device__ constexpr double kSobelV[3][3] = {{-0.125, -0.25, -0.125}, {0.0, 0.0, 0.0}, {0.125, 0.25, 0.125}};

struct Grad3 {
    double du[3], dv[3];
};

// dP/du and dP/dv at pixel (r,c); neighbours are clamped to the image (replicate padding).
__device__ inline Grad3 point_gradients(const NormalsArgs &a, const float *z, int r, int c)
{
    Grad3 g = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const int rr = min(max(r + dr, 0), a.H - 1), cc = min(max(c + dc, 0), a.W - 1);
            const double d = (double)(z[(size_t)rr * a.W + cc] + a.z_offset);  // depth + 1610 in f32 (T8:353)
            const double X = ((double)cc - a.cx) / a.fx * d;
            const double Y = ((double)rr - a.cy) / a.fy * d;
            const double ku = kSobelU[dr + 1][dc + 1], kv = kSobelV[dr + 1][dc + 1];
            g.du[0] += ku * X;
            g.du[1] += ku * Y;
            g.du[2] += ku * d;
            g.dv[0] += kv * X;
            g.dv[1] += kv * Y;
            g.dv[2] += kv * d;
        }
    }
    return g;
}

// Unit normal of pixel (r,c) as f32, y negated if requested (T8:353-354) -- shared by normals_fwd_kernel and
// the march kernel's fused epilogue so that both produce the same bits.
__device__ inline void unit_normal(const NormalsArgs &a, const float *z, int r, int c, float (&n)[3])
{
    const Grad3 g = point_gradients(a, z, r, c);
    const double nx = g.du[1] * g.dv[2] - g.du[2] * g.dv[1];
    const double ny = g.du[2] * g.dv[0] - g.du[0] * g.dv[2];
    const double nz = g.du[0] * g.dv[1] - g.du[1] * g.dv[0];
    double nn = sqrt(nx * nx + ny * ny + nz * nz);
    nn = nn > 1e-12 ? nn : 1e-12;
    n[0] = (float)(nx / nn);
    n[1] = (float)(a.negate_y ? -(ny / nn) : (ny / nn));  // T8:354
    n[2] = (float)(nz / nn);
}

// Backward of unit_normal() for one pixel: (g0,g1,g2) = dLoss/d(unit normal output, y already negated);
// scatters dLoss/d depth to the eight stencil neighbours (f32 atomics into gz, the image's grad_depth plane).
#ifndef GCFR_NBWD_INLINE
#define GCFR_NBWD_INLINE inline
#endif
__device__ GCFR_NBWD_INLINE void normals_bwd_pixel(const NormalsArgs &a, const float *z, float *gz, int r, int c,
                                         double g0, double g1_in, double g2)
{
    const Grad3 g = point_gradients(a, z, r, c);
    const double cx_ = g.du[1] * g.dv[2] - g.du[2] * g.dv[1];
    const double cy_ = g.du[2] * g.dv[0] - g.du[0] * g.dv[2];
    const double cz_ = g.du[0] * g.dv[1] - g.du[1] * g.dv[0];
    const double nrm = sqrt(cx_ * cx_ + cy_ * cy_ + cz_ * cz_);
    const double nn = nrm > 1e-12 ? nrm : 1e-12;
    const double n0 = cx_ / nn, n1 = cy_ / nn, n2 = cz_ / nn;
    const double g1 = a.negate_y ? -g1_in : g1_in;
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
