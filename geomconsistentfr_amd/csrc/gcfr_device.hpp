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

// census: @caller
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

// census: end point (T8:378-465) + ray constants
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

// census: epilogue: distance finish, tie, masked value, bonus
// sqrt(x), correctly rounded (== __builtin_sqrtf, bit for bit), for x >= 2^-96, +inf or NaN: the compiler's own IEEE expansion
// (v_sqrt_f32, then the two neighbouring floats tried with an fma residual each) WITHOUT the parts that serve inputs the march's
// epilogue cannot produce -- the 2^32 pre-scaling of denormal-range arguments and its undoing, the 0 / inf class test: 9
// instructions instead of 19.  The arguments there are S + 1e-4 and |BC|^2 + 1e-4 (T8:509, 508).  tests/test_gpu_sqrt_rn.py
// compares it with __builtin_sqrtf over EVERY float of that domain.
// (-DGCFR_R04_FIXED_COST: the A/B build of round 5's fixed-cost cuts -- the five trimmed stages in round 4's form: IEEE square roots
//  and divisions in the distance finish, the give-up test, lambert_dot() and shadow_transfer(), the 3 x 3 f64 stencil of unit_normal();
//  tools/build_variant.sh r04_fixed -DGCFR_R04_FIXED_COST, profiles/r05_fixed_cost_ab.txt)
__device__ inline float sqrt_rn_normal(float x)
{
#ifdef GCFR_R04_FIXED_COST
    return __builtin_sqrtf(x);
#endif
    const float s = __builtin_amdgcn_sqrtf(x);
    const int si = __builtin_bit_cast(int, s);
    const float s_dn = __builtin_bit_cast(float, si - 1), s_up = __builtin_bit_cast(float, si + 1);
    const float e_dn = __builtin_fmaf(-s_dn, s, x), e_up = __builtin_fmaf(-s_up, s, x);
    float o = (e_dn <= 0.0f) ? s_dn : s;
    o = (e_up > 0.0f) ? s_up : o;
    return o;
}

// census: epilogue: shading: norms (norm3_torch)
// torch's vector 2-norm accumulates acc = fma(v, v, acc) (probed; see oracle/gcfr_oracle.c).
__device__ inline float norm3_torch(float a, float b, float c)
{
    return __builtin_sqrtf(__builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a)));
}

// census: epilogue: shading: shadow transfer (expf, division; T8:517)
// T8:517: w = 1 - 4 e^-d / (1 + e^-d)^2   (== tanh^2(d/2)); evaluated as written, in f32.
// (Round 5: e^-d by v_exp_f32 on a product carried in two floats -- t = x log2(e) as hi + lo, 2^hi from the hardware, the lo part as a
//  first-order correction: ~1.5 ulp for every d, where the one-multiply __expf loses |t| ulp; libm's expf plus an IEEE division were 29
//  instructions per pixel, this is 13.  The quotient by a Newton-refined reciprocal of (1 + e)^2 in [1, 4]: within an ulp of the
//  division's.  w is compared under tolerances: <= 1e-6 against the C oracle, 2e-5 against the golden cases, gate 1e-4.)
__device__ inline float exp_neg(float d)  // e^-d, d >= 0 (NaN in, NaN out; large d: 0)
{
    const float x = -((d > 200.0f) ? 200.0f : d);  // (e^-200 is 0 in f32; keeps +inf out of the hi / lo split; NaN stays NaN)
    const float kL2eHi = 1.44269502e+0f, kL2eLo = 1.92596299e-8f;  // log2(e) = hi + lo
    const float t = x * kL2eHi;
    const float lo = __builtin_fmaf(x, kL2eLo, __builtin_fmaf(x, kL2eHi, -t));
    const float r = __builtin_amdgcn_exp2f(t);
    return __builtin_fmaf(r, lo * 0.693147182f, r);
}
__device__ inline float shadow_transfer(float d)
{
#ifdef GCFR_R04_FIXED_COST
    {
        const float e = expf(-d);
        const float onepe = 1.0f + e;
        return (-4.0f * e) / (onepe * onepe) + 1.0f;
    }
#endif
    const float e = exp_neg(d);
    const float onepe = 1.0f + e;
    const float q = onepe * onepe;
    float rq = __builtin_amdgcn_rcpf(q);
    rq = __builtin_fmaf(__builtin_fmaf(-q, rq, 1.0f), rq, rq);
    return (-4.0f * e) * rq + 1.0f;
}

// census: epilogue: shading: Lambert dot (six divisions; T8:364-366)
// Shading of one pixel for one light, T8:364-369 and 517-518 (shared by the stand-alone shade kernel
// and the fused epilogue of the march kernel so that both produce the same bits).
struct Shaded {
    float w, full, fin;
};
// n_hat . l_hat of one pixel and one light, T8:364-366 -- the forward's own arithmetic (separately rounded products and sums).
// Also called by the backward kernels where their fast evaluation of the same quantity comes out within 1e-4 of zero: the
// Lambert term max(n.l, 0) has a kink there, and which side of it a pixel is on must be the FORWARD's decision (round 3: a
// randomised soak found one pixel in 6000 cases where the multi-light backward's reciprocal-based dot had the other sign than
// the forward's -- a gradient flowed through a term the forward had clamped to zero).
// 1 / max(|v|, 1e-12) of a 3-vector (F.normalize's denominator, T8:364-365): v_rsq_f32 (1 ulp) and one Newton step instead of
// an IEEE square root (19 instructions) and, per component, an IEEE division (11): round 5's census counted 110 VALU
// instructions per pixel for the two normalisations of lambert_dot(), a seventh of the march's per-tile fixed cost.  The unit
// vector's components come out within ~1 ulp of the exactly rounded quotient (the reference's own sqrt-then-divide is within ~1
// ulp as well); shading is compared under tolerances (full_shading <= 1e-5 against the golden cases), never bit for bit.
// (|v|^2 overflowing to +inf -- |v| > 1.8e19 -- gives NaN where sqrt-then-divide gives 0; no light or normal is that long.)
__device__ inline float inv_norm3_clamped(float a, float b, float c)
{
    const float s2 = __builtin_fmaf(c, c, __builtin_fmaf(b, b, a * a));  // torch's accumulation order (norm3_torch)
    const float r = __builtin_amdgcn_rsqf(s2);
    const float r1 = r * __builtin_fmaf(-0.5f * s2 * r, r, 1.5f);  // Newton: r (3 - s2 r^2) / 2
    return s2 > 1e-24f ? r1 : 1e12f;                                 // |v| <= 1e-12 (or NaN): the clamp
}
__device__ inline float lambert_dot(float x, float y, float zb, float nx, float ny, float nz, float Cx, float Cy, float Cz)
{
    // incident light direction, T8:364
    const float lx = Cx - x, ly = Cy - y, lz = Cz - zb;
#ifdef GCFR_R04_FIXED_COST
    {
        float ln = norm3_torch(lx, ly, lz);
        ln = ln > 1e-12f ? ln : 1e-12f;
        const float ux = lx / ln, uy = ly / ln, uz = lz / ln;
        float nn = norm3_torch(nx, ny, nz);
        nn = nn > 1e-12f ? nn : 1e-12f;
        const float n0 = nx / nn, n1 = ny / nn, n2 = nz / nn;
        return (n0 * ux + n1 * uy) + n2 * uz;
    }
#endif
    const float rl = inv_norm3_clamped(lx, ly, lz);
    const float ux = lx * rl, uy = ly * rl, uz = lz * rl;
    // surface normal, re-normalised (T8:365)
    const float rn = inv_norm3_clamped(nx, ny, nz);
    const float n0 = nx * rn, n1 = ny * rn, n2 = nz * rn;
    return (n0 * ux + n1 * uy) + n2 * uz;  // T8:366
}

// census: epilogue: shading: combine (T8:366-369, 518)
__device__ inline Shaded shade_pixel(float x, float y, float zb, float nx, float ny, float nz, float Cx,
                                     float Cy, float Cz, float amb, float intensity, float min_dist)
{
    const float dot = lambert_dot(x, y, zb, nx, ny, nz, Cx, Cy, Cz);
    Shaded o;
    o.full = amb + intensity * (dot > 0.0f ? dot : 0.0f);  // T8:366-369
    o.w = shadow_transfer(min_dist);                        // T8:517
    o.fin = o.w * o.full + (1.0f - o.w) * amb;              // T8:518
    return o;
}

// census: epilogue: normals stencil (T8:353-354)
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
    double inv_fx, inv_fy;  // 1/fx, 1/fy, IEEE divisions done once on the host (set_focal)
};
inline void set_focal(NormalsArgs &a, double fx, double fy)
{
    a.fx = fx;
    a.fy = fy;
    a.inv_fx = 1.0 / fx;
    a.inv_fy = 1.0 / fy;
}

// Sobel weights (already / 8) indexed [dr+1][dc+1]
__device__ constexpr double kSobelU[3][3] = {{-0.125, 0.0, 0.125}, {-0.25, 0.0, 0.25}, {-0.125, 0.0, 0.125}};
__device__ constexpr double kSobelV[3][3] = {{-0.125, -0.25, -0.125}, {0.0, 0.0, 0.0}, {0.125, 0.25, 0.125}};

struct Grad3 {
    double du[3], dv[3];
};

// 1/x to full f64 precision (<= 1 ulp) in 5 instructions instead of the ~17 of an IEEE division: v_rcp_f64 seed (about
// 2^-26) and two Newton steps.  Normals (forward and backward) and backward kernels only -- quantities compared under
// tolerances, never bit for bit -- and only for normal, non-zero x (norms clamped to >= 1e-12, distances).
__device__ inline double fast_rcp64(double x)
{
    double r = __builtin_amdgcn_rcp(x);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    r = __builtin_fma(r, __builtin_fma(-x, r, 1.0), r);
    return r;
}

// sqrt(x) to ~1 ulp of f64 in 8 instructions instead of the ~30 of the IEEE routine: v_rsq_f64 seed and two coupled
// Newton steps (Goldschmidt).  Normals and backward kernels only, x normal and > 0 (squared norms with a floor).
__device__ inline double fast_sqrt64(double x)
{
    double y = __builtin_amdgcn_rsq(x);       // ~2^-26
    double g = x * y, h = 0.5 * y;
    double e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    h = __builtin_fma(h, e, h);
    e = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, e, g);
    return g;
}

// dP/du and dP/dv at pixel (r,c); neighbours are clamped to the image (replicate padding).
// One arithmetic for the forward (march epilogue, stand-alone kernel) and the backward's recomputation, so all of them
// see the same bits.  The reference's chain is f64 by promotion (kornia: (u - cx)/fx * d, an F.conv2d whose summation
// order is unspecified, cross, normalise) and its result is compared under a tolerance after rounding to f32; what
// matters here is instruction count -- this runs once per pixel in the march epilogue, where round 2 counted 260 f64
// instructions per pixel (six IEEE divisions, three more and an IEEE sqrt in the normalisation, 108 multiply-adds of
// which a third multiplied by the Sobel kernels' zeros): a quarter of the march's per-tile fixed cost.  Now: the
// pixel-to-ray factors are (c - cx) * (1/fx) with the reciprocal from the host, the zero taps are skipped -- except the
// centre's 0 * d, kept so that a NaN / inf depth at the pixel itself still poisons its normal as the convolution's
// zero tap does (every other neighbour has a non-zero weight in du or dv, and the cross product mixes both) -- and the
// sums are fused multiply-adds: ~115 f64 instructions, within a few ulp of f64 of the old form.
__device__ inline Grad3 point_gradients(const NormalsArgs &a, const float *z, int r, int c)
{
    Grad3 g = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    int rr[3], cc[3];
    double ax[3], ay[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        rr[j] = min(max(r + j - 1, 0), a.H - 1);
        cc[j] = min(max(c + j - 1, 0), a.W - 1);
        ax[j] = ((double)cc[j] - a.cx) * a.inv_fx;
        ay[j] = ((double)rr[j] - a.cy) * a.inv_fy;
    }
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const double d = (double)(z[(size_t)rr[dr + 1] * a.W + cc[dc + 1]] + a.z_offset);  // depth + 1610 in f32 (T8:353)
            const double ku = kSobelU[dr + 1][dc + 1], kv = kSobelV[dr + 1][dc + 1];
            if (ku == 0.0 && kv == 0.0) {  // the centre tap (compile-time after unrolling)
                g.du[2] = __builtin_fma(0.0, d, g.du[2]);
                continue;
            }
            const double X = ax[dc + 1] * d, Y = ay[dr + 1] * d;
            if (ku != 0.0) {
                g.du[0] = __builtin_fma(ku, X, g.du[0]);
                g.du[1] = __builtin_fma(ku, Y, g.du[1]);
                g.du[2] = __builtin_fma(ku, d, g.du[2]);
            }
            if (kv != 0.0) {
                g.dv[0] = __builtin_fma(kv, X, g.dv[0]);
                g.dv[1] = __builtin_fma(kv, Y, g.dv[1]);
                g.dv[2] = __builtin_fma(kv, d, g.dv[2]);
            }
        }
    }
    return g;
}

// 1/sqrt(x) to ~1 ulp of f64 in 5 instructions: v_rsq_f64 seed (about 2^-26) and one Newton step.  x normal and > 0.
__device__ inline double fast_rsqrt64(double x)
{
    const double y = __builtin_amdgcn_rsq(x);
    const double g = x * y, h = 0.5 * y;
    return __builtin_fma(y, __builtin_fma(-h, g, 0.5), y);
}

// Unit normal of pixel (r,c) as f32, y negated if requested (T8:353-354) -- shared by normals_fwd_kernel, the march
// kernel's fused epilogue and the backward kernels' recomputation of the forward's normal.
//
// Round 5 (census: 168 VALU instructions per pixel, 106 of them f64 -- a quarter of the march's per-tile fixed cost): the same
// normal from a fifth of the f64 work.  With D = depth + offset (f32, T8:353), e_ij = D_ij - D_11 the eight neighbours'
// DIFFERENCES to the centre, a = dZ/du, b = dZ/dv (Sobel / 8 of D), r = (ax, ay, 1) the pixel's own ray and ax_j = ax +- 1/fx,
// ay_i = ay +- 1/fy the neighbours' (replicate padding: the neighbour IS the pixel, "+- 0"), the point gradients are
//     dP/du = a r + p,   p = (px, py, 0),   px = [D_11 (wr + wl)/2 + (wr er + wl el)/8] / fx,   py = (wd h_2 - wu h_0) / (8 fy)
//     dP/dv = b r + q,   q = (qx, qy, 0),   qx = (wr g_2 - wl g_0) / (8 fx),   qy = [D_11 (wd + wu)/2 + (wd eb + wu et)/8] / fy
// (h_i = e_i2 - e_i0, g_j = e_2j - e_0j, er / el / eb / et the Sobel-weighted sums of the right / left column and bottom / top
// row of e, w* = 1 where that neighbour exists, 0 at the image's edge), so that
//     n = dP/du x dP/dv = r x (a q - b p) + p x q = (-my, mx, ax my - ay mx + px qy - py qx),   m = a q - b p,
// with the large common term a b (r x r) gone algebraically instead of cancelling numerically.  The differences and their small
// sums are f32: D_ij - D_11 is EXACT whenever the two are within a factor of two of each other (Sterbenz) -- every depth map
// with the reference's offset of 1610 in front of it -- and so are the sums while they stay below |D_11| / 8; everything after
// them is f64 as before.  Against the f64 restatement (oracle/normals_restatement.py): <= 0.5 f32 ulp on face-like and on
// noisy depth (amplitude 400), <= 3 ulp with an offset as small as the depth's own range; where neighbouring values differ by
// more than a factor of two the differences carry f32 rounding (relative 6e-8 of the DIFFERENCE: a perturbation of the
// surface, not of the normal's digits).  A non-finite cell poisons the same pixels as the convolution's taps do: every
// neighbour is part of a or of b, the centre of every e.
__device__ inline void unit_normal(const NormalsArgs &a, const float *z, int r, int c, float (&n)[3])
{
#ifdef GCFR_R04_FIXED_COST
    {
        const Grad3 g = point_gradients(a, z, r, c);
        const double nx = __builtin_fma(g.du[1], g.dv[2], -(g.du[2] * g.dv[1]));
        const double ny = __builtin_fma(g.du[2], g.dv[0], -(g.du[0] * g.dv[2]));
        const double nz = __builtin_fma(g.du[0], g.dv[1], -(g.du[1] * g.dv[0]));
        const double n2sum = __builtin_fma(nz, nz, __builtin_fma(ny, ny, nx * nx));
        double nn = n2sum > 1e-24 ? fast_sqrt64(n2sum) : 1e-12;
        nn = nn > 1e-12 ? nn : 1e-12;
        const double inv = fast_rcp64(nn);
        n[0] = (float)(nx * inv);
        n[1] = (float)(a.negate_y ? -(ny * inv) : (ny * inv));
        n[2] = (float)(nz * inv);
        return;
    }
#endif
    const int W = a.W;
    const bool has_l = c > 0, has_r = c < W - 1, has_u = r > 0, has_d = r < a.H - 1;
    // (raw buffer loads, byte offsets in 32 bits off one descriptor of the image's plane: no 64-bit address arithmetic per neighbour)
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(z, a.H * W * 4);
    const int b11 = (r * W + c) << 2, W4 = W << 2;
    const int bu = has_u ? b11 - W4 : b11, bd = has_d ? b11 + W4 : b11;
    const int ol = has_l ? 4 : 0, orr = has_r ? 4 : 0;
    const float off = a.z_offset;
    const float D11 = buf_load_f32(zr, b11) + off;  // depth + 1610 in f32 (T8:353)
    const float e00 = (buf_load_f32(zr, bu - ol) + off) - D11, e01 = (buf_load_f32(zr, bu) + off) - D11, e02 = (buf_load_f32(zr, bu + orr) + off) - D11;
    const float e10 = (buf_load_f32(zr, b11 - ol) + off) - D11, e12 = (buf_load_f32(zr, b11 + orr) + off) - D11;
    const float e20 = (buf_load_f32(zr, bd - ol) + off) - D11, e21 = (buf_load_f32(zr, bd) + off) - D11, e22 = (buf_load_f32(zr, bd + orr) + off) - D11;
    const float h0 = e02 - e00, h1 = e12 - e10, h2 = e22 - e20;
    const float g0 = e20 - e00, g1 = e21 - e01, g2 = e22 - e02;
    const float a8 = __builtin_fmaf(2.0f, h1, h0 + h2), b8 = __builtin_fmaf(2.0f, g1, g0 + g2);  // (2 h exact: the fma rounds once, as the sum would)
    const float er = __builtin_fmaf(2.0f, e12, e02 + e22), el = __builtin_fmaf(2.0f, e10, e00 + e20);
    const float eb = __builtin_fmaf(2.0f, e21, e20 + e22), et = __builtin_fmaf(2.0f, e01, e00 + e02);
    const float s_x = (has_r ? er : 0.0f) + (has_l ? el : 0.0f), s_y = (has_d ? eb : 0.0f) + (has_u ? et : 0.0f);
    const float v_y = (has_d ? h2 : 0.0f) - (has_u ? h0 : 0.0f), v_x = (has_r ? g2 : 0.0f) - (has_l ? g0 : 0.0f);
    const float Dx = (has_l && has_r) ? D11 : 0.5f * D11, Dy = (has_u && has_d) ? D11 : 0.5f * D11;  // (H, W >= 2: one side always exists)
    const double inv_fx = a.inv_fx, inv_fy = a.inv_fy;
    const double px = inv_fx * __builtin_fma(0.125, (double)s_x, (double)Dx), py = (0.125 * inv_fy) * (double)v_y;
    const double qx = (0.125 * inv_fx) * (double)v_x, qy = inv_fy * __builtin_fma(0.125, (double)s_y, (double)Dy);
    const double da = 0.125 * (double)a8, db = 0.125 * (double)b8;
    const double mx = __builtin_fma(da, qx, -(db * px)), my = __builtin_fma(da, qy, -(db * py));
    const double ax = ((double)c - a.cx) * inv_fx, ay = ((double)r - a.cy) * inv_fy;
    const double nx = -my, ny = mx;
    const double nz = __builtin_fma(ax, my, -(ay * mx)) + __builtin_fma(px, qy, -(py * qx));
    const double n2sum = __builtin_fma(nz, nz, __builtin_fma(ny, ny, nx * nx));
    const double inv = n2sum > 1e-24 ? fast_rsqrt64(n2sum) : 1e12;  // 1 / max(|n|, 1e-12)  (NaN: 1e12, and the products below stay NaN)
    n[0] = (float)(nx * inv);
    n[1] = (float)(a.negate_y ? -(ny * inv) : (ny * inv));  // T8:354
    n[2] = (float)(nz * inv);
}

// Backward of unit_normal() for one pixel, first half: (g0,g1,g2) = dLoss/d(unit normal output, y already negated)
// -> dLoss/d(dP/du), dLoss/d(dP/dv), the two 3-vectors every stencil neighbour's depth gradient is built from.
struct StencilGrad {
    double ddu[3], ddv[3];
};
__device__ inline StencilGrad normals_bwd_terms(const NormalsArgs &a, const float *z, int r, int c, double g0,
                                                double g1_in, double g2)
{
#pragma clang fp contract(fast)  // backward-only arithmetic: fused multiply-adds allowed (the TU default is off)
    const Grad3 g = point_gradients(a, z, r, c);
    const double cx_ = g.du[1] * g.dv[2] - g.du[2] * g.dv[1];
    const double cy_ = g.du[2] * g.dv[0] - g.du[0] * g.dv[2];
    const double cz_ = g.du[0] * g.dv[1] - g.du[1] * g.dv[0];
    const double c2sum = cx_ * cx_ + cy_ * cy_ + cz_ * cz_;
    const double nrm = c2sum > 1e-24 ? fast_sqrt64(c2sum) : 0.0;
    const double nn = nrm > 1e-12 ? nrm : 1e-12;
    const double inv_nn = fast_rcp64(nn);
    const double n0 = cx_ * inv_nn, n1 = cy_ * inv_nn, n2 = cz_ * inv_nn;
    const double g1 = a.negate_y ? -g1_in : g1_in;
    // n = c/|c|  (if |c| <= eps the denominator is the constant eps)
    double dc0, dc1, dc2;
    if (nrm > 1e-12) {
        const double ng = n0 * g0 + n1 * g1 + n2 * g2;
        dc0 = (g0 - n0 * ng) * inv_nn;
        dc1 = (g1 - n1 * ng) * inv_nn;
        dc2 = (g2 - n2 * ng) * inv_nn;
    } else {
        dc0 = g0 * inv_nn;
        dc1 = g1 * inv_nn;
        dc2 = g2 * inv_nn;
    }
    // c = du x dv:  d(du) = dv x dc,  d(dv) = dc x du
    StencilGrad o;
    o.ddu[0] = g.dv[1] * dc2 - g.dv[2] * dc1;
    o.ddu[1] = g.dv[2] * dc0 - g.dv[0] * dc2;
    o.ddu[2] = g.dv[0] * dc1 - g.dv[1] * dc0;
    o.ddv[0] = dc1 * g.du[2] - dc2 * g.du[1];
    o.ddv[1] = dc2 * g.du[0] - dc0 * g.du[2];
    o.ddv[2] = dc0 * g.du[1] - dc1 * g.du[0];
    return o;
}

// Second half, scatter form: the depth gradient of the neighbour at offset (dr, dc) of pixel (r, c) -- clamped to the
// image, i.e. replicate padding -- is  ax*dP_x + ay*dP_y + dP_z  with dP = ku*d(du) + kv*d(dv) and the neighbour's own
// (ax, ay) = ((cc - cx)/fx, (rr - cy)/fy), because P_j = (ax*d, ay*d, d).  `want(dr, dc)` selects the offsets to emit
// (f32 atomics into gz, the image's grad_depth plane).
template <class Pred>
__device__ inline void normals_bwd_scatter(const NormalsArgs &a, const StencilGrad &sg, float *gz, int r, int c, Pred want)
{
#pragma clang fp contract(fast)
    const double inv_fx = a.inv_fx, inv_fy = a.inv_fy;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
        for (int dc = -1; dc <= 1; ++dc) {
            const double ku = kSobelU[dr + 1][dc + 1], kv = kSobelV[dr + 1][dc + 1];
            if (ku == 0.0 && kv == 0.0)
                continue;
            if (!want(dr, dc))
                continue;
            const int rr = min(max(r + dr, 0), a.H - 1), cc = min(max(c + dc, 0), a.W - 1);
            const double ax = ((double)cc - a.cx) * inv_fx, ay = ((double)rr - a.cy) * inv_fy;
            const double dPx = ku * sg.ddu[0] + kv * sg.ddv[0];
            const double dPy = ku * sg.ddu[1] + kv * sg.ddv[1];
            const double dPz = ku * sg.ddu[2] + kv * sg.ddv[2];
#ifdef GCFR_ATOMIC_HOOK   // (counting build of the fused backward: gcfr_backward.hip)
            GCFR_ATOMIC_HOOK(gz + (size_t)rr * a.W + cc);
#endif
            atomicAdd(gz + (size_t)rr * a.W + cc, (float)(ax * dPx + ay * dPy + dPz));
        }
    }
}

// Backward of unit_normal() for one pixel, all eight neighbours by atomics (stand-alone kernel, multi-light kernel).
__device__ inline void normals_bwd_pixel(const NormalsArgs &a, const float *z, float *gz, int r, int c,
                                         double g0, double g1_in, double g2)
{
    const StencilGrad sg = normals_bwd_terms(a, z, r, c, g0, g1_in, g2);
    normals_bwd_scatter(a, sg, gz, r, c, [](int, int) { return true; });
}

}  // namespace gcfr
