// Backward of the render block, gfx950.
//
// The reference has no hand-written backward: torch autograd replays the ~90 tensor ops of
// train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:352-524 per image.  Only ONE sample per pixel
// carries gradient (torch.min scatters to its argmin, T8:514), so the backward of the ray march is
// a per-pixel, not a per-ray-step, computation: re-evaluate the argmin sample with the forward's
// exact position pipeline, then apply the chain rule in f64:
//     d  <- num/den,  num = sqrt(|BA x BC|^2 + 1e-4),  den = sqrt(|BC|^2 + 1e-4)
//     BA <- A - B,  A = (u_x - W/2, H/2 - u_y, zA),  zA bilinear in 4 depth corners and in u
//     u  <- start + t_k (E - start),  E <- (slope, intercept) by the selected end-point branch
//     slope, intercept <- light point C;  BC <- C - B;  B_z = own depth
// round / floor / ceil, the branch choice and the clamp are piecewise constant: zero gradient
// (exactly what autograd gives the reference).  Gradients reach: the 4 bilinear corners and the
// pixel's own depth (f32 atomics into grad_depth), and the light point (block reduction, then one
// f64 atomic per block and component).
// Counting build (-DGCFR_BWD_COUNT; tools/bwd_requests.py, round 5): the fused backward's global f32 atomics by SOURCE --
// per wave instruction: 1 instruction, the 64-B lines its active lanes touch (= requests at the memory side: each a
// read-modify-write of a line), the elements (active lanes).  Compiled out of the product.
#ifdef GCFR_BWD_COUNT
#include <hip/hip_runtime.h>
namespace gcfr {
enum { kReqOwnPixel = 0, kReqWindowFlush = 1, kReqFallbackCorners = 2, kReqStencilHalo = 3, kReqSources = 4 };
__device__ unsigned long long *g_bwd_count = nullptr;   // [source][instructions, lines, elements]
__device__ inline void bwd_count_request(int src, const void *addr)
{
    if (!g_bwd_count)
        return;
    const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
    const unsigned long long line = (unsigned long long)addr >> 6;
    unsigned long long rem = act;
    int lines = 0;
    while (rem) {
        const int l = __builtin_ctzll(rem);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)line, l), hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(line >> 32), l);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(line == (((unsigned long long)hi << 32) | lo));
        rem &= ~same;
        ++lines;
    }
    if ((int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == __builtin_ctzll(act)) {
        atomicAdd(g_bwd_count + 3 * src + 0, 1ull);
        atomicAdd(g_bwd_count + 3 * src + 1, (unsigned long long)lines);
        atomicAdd(g_bwd_count + 3 * src + 2, (unsigned long long)__builtin_popcountll(act));
    }
}
}  // namespace gcfr
#define GCFR_BWD_REQ(src, addr) gcfr::bwd_count_request(src, addr)
#define GCFR_ATOMIC_HOOK(addr) gcfr::bwd_count_request(gcfr::kReqStencilHalo, addr)
#else
#define GCFR_BWD_REQ(src, addr) ((void)0)
#endif

#include "gcfr_device.hpp"

#include <type_traits>

#include "../../include/gcfr.h"

#ifndef GCFR_BWD_INLINE
#define GCFR_BWD_INLINE inline
#endif

namespace gcfr {

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1)
        v += __shfl_down(v, off, 64);
    return v;
}

// Block-wide sum of up to NV doubles per thread -> one atomicAdd per value from thread 0.
template <int NV>
__device__ inline void block_reduce_atomic(double (&v)[NV], double *dst)
{
    __shared__ double part[4][NV];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const double s = wave_sum(v[i]);
        if (lane == 0)
            part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NV) {
        const double s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (s != 0.0)
            atomicAdd(dst + threadIdx.x, s);
    }
    __syncthreads();
}

// Wave-wide f64 sum with DPP row shifts / broadcasts (the 32-bit halves travel separately, v_add_f64 has no DPP
// form): 6 steps x (2 v_mov_dpp + 1 v_add_f64) at VALU speed instead of 12 dependent ds_bpermute round trips.
// Lanes with no source read 0 bits = +0.0.  The total ends up in lane 63.
template <int CTRL, int ROW_MASK>
__device__ inline double dpp_add_step_f64(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), CTRL, ROW_MASK, 0xf, false);
    return v + __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
__device__ inline double wave_sum_dpp(double v)
{
    v = dpp_add_step_f64<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add_step_f64<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add_step_f64<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add_step_f64<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each row holds the row's sum
    v = dpp_add_step_f64<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_add_step_f64<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's sum
    return v;
}

// Wave-wide integer minimum (DPP row shifts + row broadcasts, as the march's wave_min_i32), result in every lane via readlane.
template <int CTRL, int ROW_MASK>
__device__ inline int bwd_dpp_min_step(int v)
{
    const int moved = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false);
    return min(v, moved);
}
__device__ inline int bwd_wave_min_i32(int v)
{
    v = bwd_dpp_min_step<0x111, 0xf>(v);
    v = bwd_dpp_min_step<0x112, 0xf>(v);
    v = bwd_dpp_min_step<0x114, 0xf>(v);
    v = bwd_dpp_min_step<0x118, 0xf>(v);
    v = bwd_dpp_min_step<0x142, 0xa>(v);
    v = bwd_dpp_min_step<0x143, 0xc>(v);
    return __builtin_amdgcn_readlane(v, 63);
}

// Block-wide sum of the four per-(image, light) partials {dC.x, dC.y, dC.z, d ambient} -> one f64 atomicAdd each.
__device__ inline void block_reduce_atomic4(double (&v)[4], double *dst_light3, double *dst_ambient)
{
    __shared__ double part[4][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double s = wave_sum_dpp(v[i]);
        if (lane == 63)
            part[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double s = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
        if (s != 0.0)
            atomicAdd(threadIdx.x < 3 ? dst_light3 + threadIdx.x : dst_ambient, s);
    }
    __syncthreads();
}

// ----------------------------------------------------------------------------------------------
// shadow backward
// ----------------------------------------------------------------------------------------------
struct ShadowBwdArgs {
    const float *grad_min_dist;  // (B,L,H,W)
    const float *depth;          // (B,H,W)
    const float *light_pt;       // (B,L,3)
    const int32_t *argmin;       // (B,L,H,W)
    const double *t_table;       // (N)
    float *grad_depth;           // (B,H,W)   +=
    double *grad_light_pt;       // (B,L,3)   +=
    int32_t L, H, W, N;
};

// Backward of the ray march for ONE pixel and light: re-evaluates the argmin sample k and applies the chain
// rule (see the file header).  g32 = dLoss/d minimum_distance.  Scatters the depth gradients (five f32 atomics
// into gz, the image's grad_depth plane) and returns the light-point gradient in gC.
struct CornerScatter {  // the four bilinear-corner depth gradients of one argmin sample, to be added atomically later
    int idx[4];
    float val[4];
    int fy, fx, gy, gx;  // rows / columns of the corners (fy, fx after the -1 wrap): idx = {fy W + fx, fy W + gx, gy W + fx, gy W + gx}
};
__device__ GCFR_BWD_INLINE void shadow_bwd_pixel(const float *zimg, float *gz, const double *t_table, int H, int W,
                                        int r, int c, float Cx, float Cy, float Cz, int k, float g32,
                                        double (&gC)[3], double *own_depth_grad = nullptr,
                                        CornerScatter *defer = nullptr)
{
    const size_t p = (size_t)r * W + c;
    const Box box = image_box(H, W);
    const LightCase lc = classify_light(Cx, Cy, box);
    const double halfW = W / 2.0, halfH = H / 2.0;
    const float x = (float)c - W / 2.0f, y = H / 2.0f - (float)r;
    const float zb = zimg[p];

    // ---- forward recomputation at sample k (identical decisions to the forward kernel) ----
    const float pden = (Cx - x) + kEps4;
    const float m = (Cy - y) / pden;
    const float ic = Cy - m * Cx;
    float Ex, Ey;
    end_point(x, y, Cx, Cy, box, lc, Ex, Ey);
    // which candidate produced E, and whether the clamp cut it (zero gradient then)
    //   kind 0: constant (light inside the image)   kind 1: X candidate (xb, m*xb+ic)
    //   kind 2: Y candidate ((yb-ic)/(m+e), yb)
    int kind;
    float xb = 0.0f, yb = 0.0f;
    if (lc.xcase == 1) {
        kind = (lc.ycase == 1) ? 0 : 2;
        yb = (lc.ycase == 0) ? box.y_lo : box.y_hi;
    } else {
        xb = (lc.xcase == 0) ? box.x_lo : box.x_hi;
        if (lc.ycase == 1) {
            kind = 1;
        } else {
            yb = (lc.ycase == 0) ? box.y_lo : box.y_hi;
            const float Yx = (yb - ic) / (m + kEps4);
            kind = (Yx >= box.x_lo && Yx <= box.x_hi) ? 2 : 1;
        }
    }
    // unclamped candidate values, to detect the clamp (T8:462-465)
    float ux_raw, uy_raw;
    if (kind == 0) {
        ux_raw = Cx;
        uy_raw = Cy;
    } else if (kind == 1) {
        ux_raw = xb;
        uy_raw = m * xb + ic;
    } else {
        ux_raw = (yb - ic) / (m + kEps4);
        uy_raw = yb;
    }
    const bool live_x = !(ux_raw < box.x_lo) && !(ux_raw > box.x_hi);
    const bool live_y = !(uy_raw < box.y_lo) && !(uy_raw > box.y_hi);

    const float dxf = Ex - x, dyf = Ey - y;
    const double t = t_table[k];
    const double sx = (double)x + t * (double)dxf;
    const double sy = (double)y + t * (double)dyf;
    const double ux = (sx + halfW) - 0.0001, uy = (halfH - sy) - 0.0001;
    const double fxd = floor(ux), gxd = ceil(ux), fyd = floor(uy), gyd = ceil(uy);
    int fx = (int)fxd, gx = (int)gxd, fy = (int)fyd, gy = (int)gyd;
    fx += (fx >> 31) & W;
    fy += (fy >> 31) & H;
    const double wx0 = gxd - ux, wx1 = ux - fxd, wy0 = gyd - uy, wy1 = uy - fyd;
    const size_t iUL = (size_t)fy * W + fx, iUR = (size_t)fy * W + gx;
    const size_t iLL = (size_t)gy * W + fx, iLR = (size_t)gy * W + gx;
    const double zUL = zimg[iUL], zUR = zimg[iUR], zLL = zimg[iLL], zLR = zimg[iLR];
    const double up = zUL * wx0 + zUR * wx1, low = zLL * wx0 + zLR * wx1;
    const double zA = up * wy0 + low * wy1;
    const float Axf = (float)(ux - halfW), Ayf = (float)(halfH - uy), Azf = (float)zA;
    const double BAx = (double)(Axf - x), BAy = (double)(Ayf - y), BAz = (double)(Azf - zb);
    const double BCx = (double)(Cx - x), BCy = (double)(Cy - y), BCz = (double)(Cz - zb);
    const double Xx = BAy * BCz - BAz * BCy, Xy = BAz * BCx - BAx * BCz, Xz = BAx * BCy - BAy * BCx;
  {  // everything below is backward-only arithmetic (compared under tolerances): fused multiply-adds, fast sqrt / 1/x
#pragma clang fp contract(fast)
    const double num = fast_sqrt64(Xx * Xx + Xy * Xy + Xz * Xz + 1e-4);
    const double den = fast_sqrt64(BCx * BCx + BCy * BCy + BCz * BCz + 1e-4);

    // ---- chain rule ----
    // (reciprocal-multiply instead of IEEE divisions from here on: gradients are compared under tolerances, and an
    // f64 division costs ~17 instructions against 5 for fast_rcp64; the recomputation above stays exact because it
    // must reproduce the forward's decisions)
    const double g = (double)g32;
    const double inv_den = fast_rcp64(den), inv_num = fast_rcp64(num);
    const double dnum = g * inv_den, dden = -g * num * (inv_den * inv_den);
    const double s1 = dnum * inv_num;  // d(|X|^2+eps)^(1/2) = X/num
    const double dXx = s1 * Xx, dXy = s1 * Xy, dXz = s1 * Xz;
    const double s2 = dden * inv_den;
    double dBCx = s2 * BCx, dBCy = s2 * BCy, dBCz = s2 * BCz;
    // X = BA x BC:  dBA = BC x dX ;  dBC += dX x BA
    const double dBAx = BCy * dXz - BCz * dXy;
    const double dBAy = BCz * dXx - BCx * dXz;
    const double dBAz = BCx * dXy - BCy * dXx;
    dBCx += dXy * BAz - dXz * BAy;
    dBCy += dXz * BAx - dXx * BAz;
    dBCz += dXx * BAy - dXy * BAx;
    // B = (x, y, zb): only zb is differentiable
    const double dzb = -dBAz - dBCz;
    gC[0] = dBCx;
    gC[1] = dBCy;
    gC[2] = dBCz;
    // A = (u_x - W/2, H/2 - u_y, zA)
    const double dzA = dBAz;
    const double dzA_dux = wy0 * (zUR - zUL) + wy1 * (zLR - zLL);
    const double dzA_duy = low - up;
    const double dux = dBAx + dzA * dzA_dux;
    const double duy = -dBAy + dzA * dzA_duy;
    // u_x = s_x + W/2 - e ;  u_y = H/2 - s_y - e ;  s = start + t*(E - start)
    const double dEx = live_x ? t * dux : 0.0;
    const double dEy = live_y ? -t * duy : 0.0;
    double dm = 0.0, dic = 0.0;
    if (kind == 1) {  // E = (xb, m*xb + ic)
        dm = dEy * (double)xb;
        dic = dEy;
    } else if (kind == 2) {  // E = ((yb - ic)/(m + e), yb)
        const double inv_q = fast_rcp64((double)(m + kEps4));
        dic = -dEx * inv_q;
        dm = -dEx * (double)ux_raw * inv_q;
    }
    // ic = Cy - m*Cx ;  m = (Cy - y)/(Cx - x + e)
    gC[1] += dic;
    dm += -dic * (double)Cx;
    gC[0] += -dic * (double)m;
    const double inv_pden = fast_rcp64((double)pden);
    gC[1] += dm * inv_pden;
    gC[0] += -dm * (double)m * inv_pden;

    // depth: four bilinear corners + the pixel's own depth
    if (defer) {  // the caller issues them after its last barrier (a barrier waits for every outstanding atomic)
        defer->fy = fy;
        defer->fx = fx;
        defer->gy = gy;
        defer->gx = gx;
        defer->idx[0] = (int)iUL;
        defer->idx[1] = (int)iUR;
        defer->idx[2] = (int)iLL;
        defer->idx[3] = (int)iLR;
        defer->val[0] = (float)(dzA * wx0 * wy0);
        defer->val[1] = (float)(dzA * wx1 * wy0);
        defer->val[2] = (float)(dzA * wx0 * wy1);
        defer->val[3] = (float)(dzA * wx1 * wy1);
    } else {
        atomicAdd(gz + iUL, (float)(dzA * wx0 * wy0));
        atomicAdd(gz + iUR, (float)(dzA * wx1 * wy0));
        atomicAdd(gz + iLL, (float)(dzA * wx0 * wy1));
        atomicAdd(gz + iLR, (float)(dzA * wx1 * wy1));
    }
    if (own_depth_grad)
        *own_depth_grad += dzb;  // the caller folds it into its single own-pixel atomic
    else
        atomicAdd(gz + p, (float)dzb);
  }
}

__global__ __launch_bounds__(256) void shadow_bwd_kernel(ShadowBwdArgs a)
{
    const int H = a.H, W = a.W;
    const size_t P = (size_t)H * W;
    const int bl = blockIdx.y;
    const int b = bl / a.L;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double gC[3] = {0.0, 0.0, 0.0};

    if (p < P) {
        const int k = a.argmin[(size_t)bl * P + p];
        const float g32 = a.grad_min_dist[(size_t)bl * P + p];
        if (k >= 0 && k < a.N && g32 != 0.0f) {
            const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
            shadow_bwd_pixel(a.depth + (size_t)b * P, a.grad_depth + (size_t)b * P, a.t_table, H, W, r, c,
                             a.light_pt[3 * bl + 0], a.light_pt[3 * bl + 1], a.light_pt[3 * bl + 2], k, g32, gC);
        }
    }
    block_reduce_atomic<3>(gC, a.grad_light_pt + 3 * (size_t)bl);
}

// ----------------------------------------------------------------------------------------------
// shade backward: one thread per (image, pixel), loop over that image's lights
// ----------------------------------------------------------------------------------------------
struct ShadeBwdArgs {
    const float *normals, *depth, *albedo, *light_pt, *ambient, *min_dist;
    const float *g_w, *g_full, *g_final, *g_rendered;  // upstream grads, any may be null
    float *grad_normals;   // (B,3,H,W)  =
    float *grad_albedo;    // (B,3,H,W)  =
    float *grad_depth;     // (B,H,W)    +=
    double *grad_light_pt; // (B,L,3)    +=
    double *grad_ambient;  // (B,L)      +=
    float *grad_min_dist;  // (B,L,H,W)  =   (optional in the fused form)
    int32_t L, H, W;
    float intensity;
    // fused form (FUSED = true): normals come from the depth stencil, grad_min_dist is consumed on the spot by
    // the ray-march backward and grad_normals by the stencil backward -- one launch, no intermediate tensors
    const int32_t *argmin;       // (B,L,H,W)
    const double *t_table;       // (N)
    int32_t N;
    NormalsArgs nrm;
    const float *g_normals_out;  // (B,3,H,W) upstream grad on the returned unit normals, may be null
};

// Stand-alone shading backward (gcfr_shade_bwd: the three-kernel path), f64 chain rule, any number of lights per image;
// writes grad_normals / grad_min_dist for the stencil and march backward kernels that follow it.  (Its fused form of
// rounds 1-2 -- the march backward inlined in this loop, 200 VGPRs -- was replaced by render_bwd_multi_light_kernel.)
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void shade_bwd_kernel(ShadeBwdArgs a)
{
    const int H = a.H, W = a.W, L = a.L;
    const size_t P = (size_t)H * W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < P;
    const size_t pp = live ? p : 0;
    const int r = (int)(pp / W), c = (int)(pp - (size_t)r * W);
    const float x = (float)c - W / 2.0f, y = H / 2.0f - (float)r;
    const float *zimg = a.depth + (size_t)b * P;
    float *gz = a.grad_depth + (size_t)b * P;
    const float zb = zimg[pp];
    const double nx = a.normals[((size_t)b * 3 + 0) * P + pp];
    const double ny = a.normals[((size_t)b * 3 + 1) * P + pp];
    const double nz = a.normals[((size_t)b * 3 + 2) * P + pp];
    double nn = sqrt(nx * nx + ny * ny + nz * nz);
    nn = nn > 1e-12 ? nn : 1e-12;
    const double inv_nn = fast_rcp64(nn);
    const double n0 = nx * inv_nn, n1 = ny * inv_nn, n2 = nz * inv_nn;
    const double al0 = a.albedo[((size_t)b * 3 + 0) * P + pp];
    const double al1 = a.albedo[((size_t)b * 3 + 1) * P + pp];
    const double al2 = a.albedo[((size_t)b * 3 + 2) * P + pp];

    double gn0 = 0.0, gn1 = 0.0, gn2 = 0.0, ga0 = 0.0, ga1 = 0.0, ga2 = 0.0, gzb = 0.0;

    for (int l = 0; l < L; ++l) {
        const int bl = b * L + l;
        const size_t o = (size_t)bl * P + pp;
        const double Cx = a.light_pt[3 * bl + 0], Cy = a.light_pt[3 * bl + 1], Cz = a.light_pt[3 * bl + 2];
        const double amb = a.ambient[bl];
        double red[4] = {0.0, 0.0, 0.0, 0.0};  // dC.xyz, d ambient
        if (live) {
            // forward values
            const double lx = Cx - x, ly = Cy - y, lz = Cz - zb;
            double ln = sqrt(lx * lx + ly * ly + lz * lz);
            ln = ln > 1e-12 ? ln : 1e-12;
            const double inv_ln = fast_rcp64(ln);
            const double u0 = lx * inv_ln, u1 = ly * inv_ln, u2 = lz * inv_ln;
            double dot = n0 * u0 + n1 * u1 + n2 * u2;
            if (fabs(dot) < 1e-4)  // at the kink of max(n.l, 0) the forward's own f32 arithmetic decides the side (lambert_dot)
                dot = (double)lambert_dot(x, y, zb, (float)nx, (float)ny, (float)nz, (float)Cx, (float)Cy, (float)Cz);
            const double full = amb + (double)a.intensity * (dot > 0.0 ? dot : 0.0);
            const double e = (double)expf(-a.min_dist[o]);  // the forward evaluates the transfer function in f32 too (T8:517)
            const double ope = 1.0 + e;
            const double inv_ope = fast_rcp64(ope), inv_ope2 = inv_ope * inv_ope;
            const double w = 1.0 - 4.0 * e * inv_ope2;
            const double fin = w * full + (1.0 - w) * amb;
            // upstream
            const double gr0 = a.g_rendered ? (double)a.g_rendered[((size_t)bl * 3 + 0) * P + pp] : 0.0;
            const double gr1 = a.g_rendered ? (double)a.g_rendered[((size_t)bl * 3 + 1) * P + pp] : 0.0;
            const double gr2 = a.g_rendered ? (double)a.g_rendered[((size_t)bl * 3 + 2) * P + pp] : 0.0;
            ga0 += gr0 * fin;  // rendered = albedo * final  (T8:520-522)
            ga1 += gr1 * fin;
            ga2 += gr2 * fin;
            const double dfin = (gr0 * al0 + gr1 * al1 + gr2 * al2) + (a.g_final ? (double)a.g_final[o] : 0.0);
            // final = w*full + (1-w)*amb  (T8:518)
            const double dw = dfin * (full - amb) + (a.g_w ? (double)a.g_w[o] : 0.0);
            const double dfull = dfin * w + (a.g_full ? (double)a.g_full[o] : 0.0);
            red[3] = dfin * (1.0 - w) + dfull;  // ambient enters final directly and through full
            // w = 1 - 4e/(1+e)^2, e = exp(-d):  dw/dd = 4e(1-e)/(1+e)^3  (T8:517)
            const float gmd = (float)(dw * (4.0 * e * (1.0 - e)) * (inv_ope2 * inv_ope));
            if (a.grad_min_dist)
                a.grad_min_dist[o] = gmd;
            // full = amb + I*max(dot,0)  (T8:366)
            const double ddot = (dot > 0.0) ? dfull * (double)a.intensity : 0.0;
            const double dn0 = ddot * u0, dn1 = ddot * u1, dn2 = ddot * u2;  // d n_hat
            const double du0 = ddot * n0, du1 = ddot * n1, du2 = ddot * n2;  // d l_hat
            // n_hat = n/|n|
            const double nd = n0 * dn0 + n1 * dn1 + n2 * dn2;
            gn0 += (dn0 - n0 * nd) * inv_nn;
            gn1 += (dn1 - n1 * nd) * inv_nn;
            gn2 += (dn2 - n2 * nd) * inv_nn;
            // l_hat = l/|l|, l = C - P
            const double ud = u0 * du0 + u1 * du1 + u2 * du2;
            const double dl0 = (du0 - u0 * ud) * inv_ln, dl1 = (du1 - u1 * ud) * inv_ln, dl2 = (du2 - u2 * ud) * inv_ln;
            red[0] += dl0;
            red[1] += dl1;
            red[2] += dl2;
            gzb -= dl2;
        }
        // per-(image, light) reductions: light point (3) and ambient (1), one pass through LDS
        block_reduce_atomic4(red, a.grad_light_pt + 3 * (size_t)bl, a.grad_ambient + bl);
    }
    if (live) {
        a.grad_normals[((size_t)b * 3 + 0) * P + p] = (float)gn0;
        a.grad_normals[((size_t)b * 3 + 1) * P + p] = (float)gn1;
        a.grad_normals[((size_t)b * 3 + 2) * P + p] = (float)gn2;
        a.grad_albedo[((size_t)b * 3 + 0) * P + p] = (float)ga0;
        a.grad_albedo[((size_t)b * 3 + 1) * P + p] = (float)ga1;
        a.grad_albedo[((size_t)b * 3 + 2) * P + p] = (float)ga2;
        atomicAdd(gz + p, (float)gzb);
    }
}

// ----------------------------------------------------------------------------------------------
// fused backward, one light per image (the training shape: T8 has one predicted light per face)
//
// Same per-pixel device functions as the stand-alone kernels, but staged so that each phase
// keeps only its own operands alive: (1) shading backward -- writes grad_albedo at once and leaves three f32 normal
// gradients, the own-depth term, the four reduction partials and the f32 gradient on the minimum distance;
// (2) ray-march backward through the argmin sample; (3) block reduction; (4) normals-stencil backward.  The
// general kernel runs (1)-(2) inside a loop over lights, which keeps every loop-invariant (unit normal, albedo,
// accumulators in f64) alive across the inlined march backward: 232 VGPRs, two waves per SIMD.
// ----------------------------------------------------------------------------------------------
#ifdef GCFR_BWD_TRACE   // debug build: per-wave stage stamps of render_bwd_single_light_kernel (tools/bwd_stage_trace.py)
__device__ unsigned long long *g_bwd_trace = nullptr;
#define GCFR_BWD_STAMP(i) (stamp[i] = __builtin_amdgcn_s_memrealtime())
#else
#define GCFR_BWD_STAMP(i) ((void)0)
#endif

#ifndef GCFR_BWD1_WAVES_PER_EU
#define GCFR_BWD1_WAVES_PER_EU 4
#endif
constexpr int kBwdTileW = 32, kBwdTileH = 8;  // 256 threads
#ifndef GCFR_BWD_WINDOW
#define GCFR_BWD_WINDOW 1
#endif
constexpr int kBwdWinW = 64, kBwdWinH = 48;  // the corner window: 64 columns (one lane each at the flush) x 48 rows of f32 = 12 KiB
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_BWD1_WAVES_PER_EU, GCFR_BWD1_WAVES_PER_EU))) void render_bwd_single_light_kernel(ShadeBwdArgs a)
{
    const int H = a.H, W = a.W;
    const size_t P = (size_t)H * W;
    const int b = blockIdx.y;  // == the (image, light) index: L = 1
    // One workgroup marches over several kBwdTileW x kBwdTileH pixel tiles (rows of 32 f32 = whole 128-B lines) of ONE
    // image: tiles, so that the stencil backward can exchange most of its eight-neighbour contributions through LDS
    // instead of global atomics; several, so that the per-image light / ambient gradients (four f64 atomics per
    // workgroup, all workgroups of an image on the same two cache lines, ~20 ns each when they queue on one line)
    // are issued once per `tiles per workgroup` tiles -- with one tile per workgroup that queue alone was 80 us long.
    const int tiles_x = (W + kBwdTileW - 1) / kBwdTileW;
    const int n_tiles = tiles_x * ((H + kBwdTileH - 1) / kBwdTileH);
    const int ty = (int)threadIdx.x / kBwdTileW, tx = (int)threadIdx.x - ty * kBwdTileW;
    const float *zimg = a.depth + (size_t)b * P;
    float *gz = a.grad_depth + (size_t)b * P;
    const float Cxf = a.light_pt[3 * b + 0], Cyf = a.light_pt[3 * b + 1], Czf = a.light_pt[3 * b + 2];
    __shared__ double s_dd[6][256];
    __shared__ double s_part[4][4];
    double red[4] = {0.0, 0.0, 0.0, 0.0};  // dC.xyz, d ambient: accumulated over this workgroup's tiles
#if GCFR_BWD_WINDOW
    // Corner window (round 4).  The four bilinear-corner gradients of a tile's 256 argmin samples are what the memory system
    // spends this kernel's time on: every atomic REQUEST (one per wave instruction and 64-B line touched) is a read-modify-write
    // at the memory side.  Neighbouring pixels' samples land on neighbouring -- often the same -- texels (parallel rays, similar
    // argmin fractions), so a tile's corners usually fall inside a small rectangle of the image: they are first summed there,
    // in LDS (ds_add_f32), and the rectangle is then flushed row by row, one lane per column -- whole 64-B lines per request,
    // each texel at most once per tile.  A tile whose corners do not fit the window (scattered argmins: rough depth; the -1
    // wrap) falls back to the register run-merge + direct atomics below.
    __shared__ float s_win[kBwdWinH * kBwdWinW];
    __shared__ int s_box[4][4];
    for (int i = (int)threadIdx.x; i < kBwdWinH * kBwdWinW; i += 256)
        s_win[i] = 0.0f;   // (every flush leaves the cells it read at zero again)
#endif

#ifdef GCFR_BWD_TRACE
    unsigned long long stamp[8] = {};
    unsigned long long acc[8] = {};
#endif
  for (int tile = (int)blockIdx.x; tile < n_tiles; tile += (int)gridDim.x) {
    GCFR_BWD_STAMP(0);
    const int tile_r = tile / tiles_x, tile_c = tile - tile_r * tiles_x;
    const int r_raw = tile_r * kBwdTileH + ty, c_raw = tile_c * kBwdTileW + tx;
    const bool live = (r_raw < H) && (c_raw < W);
    const int r = live ? r_raw : 0, c = live ? c_raw : 0;
    const size_t pp = (size_t)r * W + c, p = pp;
    float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;  // gradient on the unit normal, rounded to f32 as the unfused path stores it
    float gmd = 0.0f;
    double gzb = 0.0;
    if (live) {  // ---- (1) shading backward ----
#pragma clang fp contract(fast)  // backward-only arithmetic (compared under tolerances)
        // f32: the forward evaluates this stage in f32 (shade_pixel), and every result leaves it as f32 (grad_albedo,
        // the normal gradient handed to the stencil, the gradient on the minimum distance) or enters an f64 sum
        // over 65 536 pixels (light, ambient); an f64 VALU op costs 1.6x an f32 one here and twice the registers.
        const size_t o = (size_t)b * P + pp;
        const float gr0 = a.g_rendered ? a.g_rendered[((size_t)b * 3 + 0) * P + pp] : 0.0f;
        const float gr1 = a.g_rendered ? a.g_rendered[((size_t)b * 3 + 1) * P + pp] : 0.0f;
        const float gr2 = a.g_rendered ? a.g_rendered[((size_t)b * 3 + 2) * P + pp] : 0.0f;
        const float gfin = a.g_final ? a.g_final[o] : 0.0f, gw = a.g_w ? a.g_w[o] : 0.0f, gfull = a.g_full ? a.g_full[o] : 0.0f;
        // A pixel nothing upstream depends on (the training losses mask the rendered image: about half of a face
        // batch) contributes exactly zero to every gradient of this stage and of the march backward.
        if (gr0 != 0.0f || gr1 != 0.0f || gr2 != 0.0f || gfin != 0.0f || gw != 0.0f || gfull != 0.0f) {
            const float x = (float)c - W / 2.0f, y = H / 2.0f - (float)r;
            const float zb = zimg[pp];
            float n[3];
            if (a.normals) {  // the unit normals the forward wrote (normals_out): no second stencil evaluation
                n[0] = a.normals[((size_t)b * 3 + 0) * P + pp];
                n[1] = a.normals[((size_t)b * 3 + 1) * P + pp];
                n[2] = a.normals[((size_t)b * 3 + 2) * P + pp];
            } else {
                unit_normal(a.nrm, zimg, r, c, n);
            }
            float nn = __builtin_sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
            nn = nn > 1e-12f ? nn : 1e-12f;
            const float inv_nn = 1.0f / nn;
            const float n0 = n[0] * inv_nn, n1 = n[1] * inv_nn, n2 = n[2] * inv_nn;
            const float amb = a.ambient[b];
            const float lx = Cxf - x, ly = Cyf - y, lz = Czf - zb;
            float ln = __builtin_sqrtf(lx * lx + ly * ly + lz * lz);
            ln = ln > 1e-12f ? ln : 1e-12f;
            const float inv_ln = 1.0f / ln;
            const float u0 = lx * inv_ln, u1 = ly * inv_ln, u2 = lz * inv_ln;
            float dot = n0 * u0 + n1 * u1 + n2 * u2;
            if (fabsf(dot) < 1e-4f)  // at the kink of max(n.l, 0) the forward's own arithmetic decides the side (lambert_dot)
                dot = lambert_dot(x, y, zb, n[0], n[1], n[2], Cxf, Cyf, Czf);
            const float full = amb + a.intensity * (dot > 0.0f ? dot : 0.0f);
            const float e = expf(-a.min_dist[o]);  // T8:517
            const float ope = 1.0f + e;
            const float inv_ope = 1.0f / ope, inv_ope2 = inv_ope * inv_ope;
            const float w = 1.0f - 4.0f * e * inv_ope2;
            const float fin = w * full + (1.0f - w) * amb;
            const float al0 = a.albedo[((size_t)b * 3 + 0) * P + pp];
            const float al1 = a.albedo[((size_t)b * 3 + 1) * P + pp];
            const float al2 = a.albedo[((size_t)b * 3 + 2) * P + pp];
            a.grad_albedo[((size_t)b * 3 + 0) * P + p] = gr0 * fin;  // rendered = albedo * final  (T8:520-522)
            a.grad_albedo[((size_t)b * 3 + 1) * P + p] = gr1 * fin;
            a.grad_albedo[((size_t)b * 3 + 2) * P + p] = gr2 * fin;
            const float dfin = (gr0 * al0 + gr1 * al1 + gr2 * al2) + gfin;
            const float dw = dfin * (full - amb) + gw;  // final = w*full + (1-w)*amb (T8:518)
            const float dfull = dfin * w + gfull;
            red[3] += (double)(dfin * (1.0f - w) + dfull);  // ambient enters final directly and through full
            gmd = dw * (4.0f * e * (1.0f - e)) * (inv_ope2 * inv_ope);  // dw/dd = 4e(1-e)/(1+e)^3 (T8:517)
            const float ddot = (dot > 0.0f) ? dfull * a.intensity : 0.0f;  // full = amb + I*max(dot,0) (T8:366)
            const float dn0 = ddot * u0, dn1 = ddot * u1, dn2 = ddot * u2;
            const float du0 = ddot * n0, du1 = ddot * n1, du2 = ddot * n2;
            const float nd = n0 * dn0 + n1 * dn1 + n2 * dn2;  // n_hat = n/|n|
            h0 = (dn0 - n0 * nd) * inv_nn;
            h1 = (dn1 - n1 * nd) * inv_nn;
            h2 = (dn2 - n2 * nd) * inv_nn;
            const float ud = u0 * du0 + u1 * du1 + u2 * du2;  // l_hat = l/|l|, l = C - P
            const float dl2 = (du2 - u2 * ud) * inv_ln;
            red[0] += (double)((du0 - u0 * ud) * inv_ln);
            red[1] += (double)((du1 - u1 * ud) * inv_ln);
            red[2] += (double)dl2;
            gzb = -(double)dl2;
        } else {
            a.grad_albedo[((size_t)b * 3 + 0) * P + p] = 0.0f;
            a.grad_albedo[((size_t)b * 3 + 1) * P + p] = 0.0f;
            a.grad_albedo[((size_t)b * 3 + 2) * P + p] = 0.0f;
        }
    }
    GCFR_BWD_STAMP(1);
    CornerScatter corners = {{0, 0, 0, 0}, {0.0f, 0.0f, 0.0f, 0.0f}};
    bool have_corners = false;
    if (live) {  // ---- (2) ray-march backward through the argmin sample (atomics deferred) ----
        const int k = a.argmin[(size_t)b * P + pp];
        if (k >= 0 && k < a.N && gmd != 0.0f) {
            double gC[3];
            shadow_bwd_pixel(zimg, gz, a.t_table, H, W, r, c, Cxf, Cyf, Czf, k, gmd, gC, &gzb, &corners);
            have_corners = true;
            red[0] += gC[0];
            red[1] += gC[1];
            red[2] += gC[2];
        }
    }
    GCFR_BWD_STAMP(2);
    // ---- (4) stencil backward: ONE barrier per tile, no global atomic in front of it (a barrier waits for every
    // outstanding memory operation of the wave -- scattered atomics before it would be paid as latency by the whole
    // workgroup).  The L2 executes f32 atomic elements at a finite rate and the scatter form costs 14 per pixel (8
    // stencil neighbours, 4 bilinear corners, own pixel twice): gather form instead -- every pixel publishes its two
    // stencil vectors in LDS, sums what its in-tile neighbours owe it and issues ONE atomic for its own depth; only
    // contributions that cross the tile border (0.9 per pixel) and those of image-border pixels (replicate padding
    // folds several offsets onto one target) remain scattered.
    StencilGrad sg = {{0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
    bool have_sg = false;
    const bool on_image_border = (r == 0) || (c == 0) || (r == H - 1) || (c == W - 1);
    if (live) {
        double g0 = (double)h0, g1 = (double)h1, g2 = (double)h2;
        if (a.g_normals_out) {
            g0 += a.g_normals_out[((size_t)b * 3 + 0) * P + p];
            g1 += a.g_normals_out[((size_t)b * 3 + 1) * P + p];
            g2 += a.g_normals_out[((size_t)b * 3 + 2) * P + p];
        }
        if (g0 != 0.0 || g1 != 0.0 || g2 != 0.0) {  // (zero in, zero out: masked-out pixels skip the stencil)
            sg = normals_bwd_terms(a.nrm, zimg, r, c, g0, g1, g2);
            have_sg = true;
        }
    }
    const bool via_lds = live && !on_image_border;
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        s_dd[q][threadIdx.x] = via_lds ? sg.ddu[q] : 0.0;
        s_dd[3 + q][threadIdx.x] = via_lds ? sg.ddv[q] : 0.0;
    }
#if GCFR_BWD_WINDOW
    {   // this wave's corner box {r_min, c_min, -r_max, -c_max}; lanes without a sample do not count
        const bool hc = live && have_corners;
        const int big = 0x3fffffff;
        const int wr0 = bwd_wave_min_i32(hc ? min(corners.fy, corners.gy) : big), wc0 = bwd_wave_min_i32(hc ? min(corners.fx, corners.gx) : big);
        const int wr1 = bwd_wave_min_i32(hc ? -max(corners.fy, corners.gy) : big), wc1 = bwd_wave_min_i32(hc ? -max(corners.fx, corners.gx) : big);
        if ((threadIdx.x & 63) == 0) {
            s_box[threadIdx.x >> 6][0] = wr0;
            s_box[threadIdx.x >> 6][1] = wc0;
            s_box[threadIdx.x >> 6][2] = wr1;
            s_box[threadIdx.x >> 6][3] = wc1;
        }
    }
#endif
    GCFR_BWD_STAMP(3);
    __syncthreads();
    GCFR_BWD_STAMP(4);
    if (live) {
        // what the in-tile neighbours owe this pixel: source = this pixel minus the offset
        double Sx = 0.0, Sy = 0.0, Sz = 0.0;
#pragma unroll
        for (int dr = -1; dr <= 1; ++dr) {
#pragma unroll
            for (int dc = -1; dc <= 1; ++dc) {
                const double ku = kSobelU[dr + 1][dc + 1], kv = kSobelV[dr + 1][dc + 1];
                if (ku == 0.0 && kv == 0.0)
                    continue;
                const int sy = ty - dr, sx = tx - dc;
                if ((unsigned)sy < (unsigned)kBwdTileH && (unsigned)sx < (unsigned)kBwdTileW) {
                    const int src = sy * kBwdTileW + sx;  // (a source on the image border or outside it published zeros)
                    Sx += ku * s_dd[0][src] + kv * s_dd[3][src];
                    Sy += ku * s_dd[1][src] + kv * s_dd[4][src];
                    Sz += ku * s_dd[2][src] + kv * s_dd[5][src];
                }
            }
        }
        const double ax = ((double)c - a.nrm.cx) * a.nrm.inv_fx, ay = ((double)r - a.nrm.cy) * a.nrm.inv_fy;
        GCFR_BWD_REQ(kReqOwnPixel, gz + p);
        atomicAdd(gz + p, (float)((ax * Sx + ay * Sy + Sz) + gzb));
    }
#if GCFR_BWD_WINDOW
    // (wave-uniform: every thread reduces the four waves' boxes itself)
    const int box_r0 = min(min(s_box[0][0], s_box[1][0]), min(s_box[2][0], s_box[3][0]));
    const int box_c0 = min(min(s_box[0][1], s_box[1][1]), min(s_box[2][1], s_box[3][1]));
    const int box_r1 = -min(min(s_box[0][2], s_box[1][2]), min(s_box[2][2], s_box[3][2]));
    const int box_c1 = -min(min(s_box[0][3], s_box[1][3]), min(s_box[2][3], s_box[3][3]));
    const bool any_corner = box_r0 <= box_r1;
    const bool use_window = any_corner && (box_r1 - box_r0 < kBwdWinH) && (box_c1 - box_c0 < kBwdWinW);
    if (use_window) {
        if (live && have_corners) {
            const int o00 = (corners.fy - box_r0) * kBwdWinW + (corners.fx - box_c0), dgy = (corners.gy - corners.fy) * kBwdWinW,
                      dgx = corners.gx - corners.fx;
            __hip_atomic_fetch_add(&s_win[o00], corners.val[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_win[o00 + dgx], corners.val[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_win[o00 + dgy], corners.val[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_fetch_add(&s_win[o00 + dgy + dgx], corners.val[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __syncthreads();
        {   // flush: wave w takes rows w, w + 4, ...; lane = column
            const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
            const int nrows = box_r1 - box_r0 + 1, ncols = box_c1 - box_c0 + 1;
            for (int rw = wave; rw < nrows; rw += 4) {
                const float v = s_win[rw * kBwdWinW + lane];
                if (lane < ncols) {
                    // (every cell read is zeroed, not only the ones flushed: a denormal sum compares equal to zero under
                    //  flush-to-zero and would otherwise stay behind for the next tile's window -- advisor r04)
                    s_win[rw * kBwdWinW + lane] = 0.0f;
                    if (v != 0.0f) {
                        GCFR_BWD_REQ(kReqWindowFlush, gz + (size_t)(box_r0 + rw) * W + (box_c0 + lane));
                        atomicAdd(gz + (size_t)(box_r0 + rw) * W + (box_c0 + lane), v);
                    }
                }
            }
        }
    } else
#endif
    if (true) {
    // The four bilinear-corner atomics of the argmin samples were 78 of this kernel's 167 us: neighbouring pixels march
    // nearly parallel rays, so at sample fraction t their samples are only (1 - t) texels apart and 2 ... 5 adjacent lanes
    // hit the SAME texel -- the L2 serialises those.  Runs of adjacent lanes (inside aligned groups of 8) with identical
    // corner addresses are therefore summed in registers first (segmented scan over DPP row shifts, 3 steps) and only the
    // last lane of a run issues the atomics.  Wave-uniform code: every lane takes part, lanes without a sample form runs
    // of their own.
    {
        const int lane = threadIdx.x & 63;
        // Run identity: idx[0] plus three FLAGS that fix the other corners given idx[0] -- right neighbour differs, it is the
        // wrapped column (idx[1] < idx[0]: fx = -1 -> W-1, gx = 0; T8:488-491), lower neighbour differs (fy = -1 wraps the
        // same way, and then gy = 0 is implied by idx[0]'s row).  Round 2 packed the DIFFERENCE idx[1] - idx[0] into the key
        // as if it were 0 or 1: on the wrapped column it is -(W-1), which set the sign bit for W % 16 in {2,4,6,8} and
        // dropped the run's atomics (advisor, r02).  H, W <= 4096 (checked by the entry point), so idx[0] < 2^24.
        // (bit 31 stays clear: key >= 0 <=> the lane has a sample)
        const int kx = corners.idx[1] - corners.idx[0];
        const int flags = ((kx != 0) ? (1 << 28) : 0) | ((kx < 0) ? (1 << 30) : 0) | ((corners.idx[2] != corners.idx[0]) ? (1 << 29) : 0);
        const int key = (live && have_corners) ? (corners.idx[0] | flags) : (-1 - lane);   // lanes without a sample: unique, never equal
        const int key_prev = __builtin_amdgcn_update_dpp(0, key, 0x111, 0xf, 0xf, true);  // row_shr:1
        int head = ((lane & 7) == 0) || (key_prev != key);
        float v0 = corners.val[0], v1 = corners.val[1], v2 = corners.val[2], v3 = corners.val[3];
        auto shr = [](float v, auto ctrl) {
            return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), decltype(ctrl)::value, 0xf, 0xf, true));
        };
        auto step = [&](auto ctrl, int d) {
            const float a0 = shr(v0, ctrl), a1 = shr(v1, ctrl), a2 = shr(v2, ctrl), a3 = shr(v3, ctrl);
            const int hp = __builtin_amdgcn_update_dpp(1, head, decltype(ctrl)::value, 0xf, 0xf, false);
            const bool take = ((lane & 7) >= d) && !head;
            v0 = take ? v0 + a0 : v0;
            v1 = take ? v1 + a1 : v1;
            v2 = take ? v2 + a2 : v2;
            v3 = take ? v3 + a3 : v3;
            head = take ? (head | hp) : head;
        };
        step(std::integral_constant<int, 0x111>{}, 1);  // row_shr:1
        step(std::integral_constant<int, 0x112>{}, 2);  // row_shr:2
        step(std::integral_constant<int, 0x114>{}, 4);  // row_shr:4
        // the run ends here if the next lane starts a new one (computed from the ORIGINAL keys)
        const int key_next = __builtin_amdgcn_update_dpp(0, key, 0x101, 0xf, 0xf, true);   // row_shl:1
        const bool last = ((lane & 7) == 7) || (key_next != key);
        if (last && key >= 0) {
            GCFR_BWD_REQ(kReqFallbackCorners, gz + corners.idx[0]);
            GCFR_BWD_REQ(kReqFallbackCorners, gz + corners.idx[1]);
            GCFR_BWD_REQ(kReqFallbackCorners, gz + corners.idx[2]);
            GCFR_BWD_REQ(kReqFallbackCorners, gz + corners.idx[3]);
            atomicAdd(gz + corners.idx[0], v0);
            atomicAdd(gz + corners.idx[1], v1);
            atomicAdd(gz + corners.idx[2], v2);
            atomicAdd(gz + corners.idx[3], v3);
        }
    }

    }
    if (live) {
        // (Round 3 measured gathering the tile's 84 halo pixels as well -- one atomic per halo pixel instead of up to five
        //  per edge pixel, 0.9 -> 0.33 atomic elements per pixel: 145 -> 138 us on a dense upstream gradient, but 93 -> 96 us
        //  inside the training step, whose masked gradient leaves half the tiles without stencil work, and 20 B more
        //  scratch; HBM traffic unchanged at 293-300 MB.  Not kept: profiles/r03_backward_ab.txt.)
        if (!have_sg) {
            // nothing to scatter
        } else if (on_image_border)  // clamped targets: all eight by atomics, as the stand-alone kernel does
            normals_bwd_scatter(a.nrm, sg, gz, r, c, [](int, int) { return true; });
        else                  // targets outside this tile only
            normals_bwd_scatter(a.nrm, sg, gz, r, c, [&](int dr, int dc) {
                return (unsigned)(ty + dr) >= (unsigned)kBwdTileH || (unsigned)(tx + dc) >= (unsigned)kBwdTileW;
            });
    }
    GCFR_BWD_STAMP(5);
    __syncthreads();  // s_dd is rewritten by the next tile
    GCFR_BWD_STAMP(6);
#ifdef GCFR_BWD_TRACE
    for (int i = 0; i < 6; ++i)
        acc[i] += stamp[i + 1] - stamp[i];
    acc[6] += 1;
#endif
  }
    // ---- (3) per-image reductions: light point (3) and ambient (1), once per workgroup ----
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double ws = wave_sum_dpp(red[i]);
        if ((threadIdx.x & 63) == 63)
            s_part[threadIdx.x >> 6][i] = ws;
    }
    __syncthreads();
    if (threadIdx.x < 4) {
        const double sum = (s_part[0][threadIdx.x] + s_part[1][threadIdx.x]) + (s_part[2][threadIdx.x] + s_part[3][threadIdx.x]);
        if (sum != 0.0)
            atomicAdd(threadIdx.x < 3 ? a.grad_light_pt + 3 * (size_t)b + threadIdx.x : a.grad_ambient + b, sum);
    }
#ifdef GCFR_BWD_TRACE
    if (g_bwd_trace && (threadIdx.x & 63) == 0) {
        unsigned long long *rec = g_bwd_trace + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8;
        for (int i = 0; i < 7; ++i)
            rec[i] = acc[i];
        rec[7] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ----------------------------------------------------------------------------------------------
// light-prep backward (n tiny): unit = l'/max(|l'|,eps), l' = (a, b, max(c, clamp)), pt = dist*unit
// ----------------------------------------------------------------------------------------------
__global__ void light_prep_bwd_kernel(const float *__restrict__ light_raw, int n, int clamp_z,
                                      float clamp_min, float light_distance,
                                      const float *__restrict__ grad_unit,
                                      const double *__restrict__ grad_light_pt,
                                      float *__restrict__ grad_light_raw)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const double a = light_raw[3 * i + 0], b = light_raw[3 * i + 1];
    double c = light_raw[3 * i + 2];
    double zpass = 1.0;
    if (clamp_z) {  // torch.maximum backward: 1 above, 0 below, 1/2 on a tie
        zpass = (c > (double)clamp_min) ? 1.0 : (c == (double)clamp_min ? 0.5 : 0.0);
        c = (c > (double)clamp_min) ? c : (double)clamp_min;
    }
    const double nrm = sqrt(a * a + b * b + c * c);
    const double d = nrm > 1e-12 ? nrm : 1e-12;
    const double u0 = a / d, u1 = b / d, u2 = c / d;
    double g0 = grad_light_pt ? (double)light_distance * grad_light_pt[3 * i + 0] : 0.0;
    double g1 = grad_light_pt ? (double)light_distance * grad_light_pt[3 * i + 1] : 0.0;
    double g2 = grad_light_pt ? (double)light_distance * grad_light_pt[3 * i + 2] : 0.0;
    if (grad_unit) {
        g0 += grad_unit[3 * i + 0];
        g1 += grad_unit[3 * i + 1];
        g2 += grad_unit[3 * i + 2];
    }
    double r0, r1, r2;
    if (nrm > 1e-12) {
        const double ug = u0 * g0 + u1 * g1 + u2 * g2;
        r0 = (g0 - u0 * ug) / d;
        r1 = (g1 - u1 * ug) / d;
        r2 = (g2 - u2 * ug) / d;
    } else {  // clamp_min(eps) active: the denominator is a constant
        r0 = g0 / d;
        r1 = g1 / d;
        r2 = g2 / d;
    }
    grad_light_raw[3 * i + 0] = (float)r0;
    grad_light_raw[3 * i + 1] = (float)r1;
    grad_light_raw[3 * i + 2] = (float)(r2 * zpass);
}

// ----------------------------------------------------------------------------------------------
// fused backward, several lights per image (BASELINE configs[4]: 18 lights per face), restaged in round 3
//
// Round 1's fused kernel (shade_bwd_kernel<true>) runs its shading backward in f64 inside the loop over lights and keeps
// every loop-invariant -- unit normal, albedo, six f64 accumulators -- alive across the inlined march backward: 200
// VGPRs, two waves per SIMD.  This kernel is the single-light kernel's staging applied per light: stage (1) in f32
// exactly as render_bwd_single_light_kernel evaluates it (so L = 1 and L > 1 give the same numbers), its operands
// RE-READ per light (normal, albedo: three cached loads each, instead of 14 registers across the march backward), and
// only seven f32 / one f64 accumulators carried through stage (2).  One lane per pixel, 256-pixel blocks; the corner
// atomics go out at once and the stencil backward scatters (8 atomics per pixel, once per pixel whatever L).
// 144 VGPRs, three waves per SIMD, no scratch (round 1's kernel: 200 VGPRs, two waves; forcing it to four spilled 268 B).
// ----------------------------------------------------------------------------------------------
#ifndef GCFR_BWDL_WAVES_PER_EU
#define GCFR_BWDL_WAVES_PER_EU 3
#endif
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_BWDL_WAVES_PER_EU, GCFR_BWDL_WAVES_PER_EU))) void render_bwd_multi_light_kernel(ShadeBwdArgs a)
{
    const int H = a.H, W = a.W, L = a.L;
    const size_t P = (size_t)H * W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = p < P;
    const size_t pp = live ? p : 0;
    const int r = (int)(pp / W), c = (int)(pp - (size_t)r * W);
    const float *zimg = a.depth + (size_t)b * P;
    float *gz = a.grad_depth + (size_t)b * P;
    float h0 = 0.0f, h1 = 0.0f, h2 = 0.0f;     // gradient on the unit normal, summed over the lights
    float ga0 = 0.0f, ga1 = 0.0f, ga2 = 0.0f;  // gradient on the albedo
    double gzb = 0.0;                          // own-depth terms of all stages
    float npre[3] = {0.0f, 0.0f, 1.0f};        // without the forward's normals: the stencil is evaluated ONCE, in front of the loop
    if (live && !a.normals)
        unit_normal(a.nrm, zimg, r, c, npre);
#pragma clang loop unroll(disable)
    for (int l = 0; l < L; ++l) {
        const int bl = b * L + l;
        const float Cxf = a.light_pt[3 * bl + 0], Cyf = a.light_pt[3 * bl + 1], Czf = a.light_pt[3 * bl + 2];
        double red[4] = {0.0, 0.0, 0.0, 0.0};  // dC.xyz, d ambient of this light
        float gmd = 0.0f;
        if (live) {  // ---- (1) shading backward, f32 as the forward (see render_bwd_single_light_kernel) ----
#pragma clang fp contract(fast)
            const size_t o = (size_t)bl * P + pp;
            const float gr0 = a.g_rendered ? a.g_rendered[((size_t)bl * 3 + 0) * P + pp] : 0.0f;
            const float gr1 = a.g_rendered ? a.g_rendered[((size_t)bl * 3 + 1) * P + pp] : 0.0f;
            const float gr2 = a.g_rendered ? a.g_rendered[((size_t)bl * 3 + 2) * P + pp] : 0.0f;
            const float gfin = a.g_final ? a.g_final[o] : 0.0f, gw = a.g_w ? a.g_w[o] : 0.0f, gfull = a.g_full ? a.g_full[o] : 0.0f;
            if (gr0 != 0.0f || gr1 != 0.0f || gr2 != 0.0f || gfin != 0.0f || gw != 0.0f || gfull != 0.0f) {
                const float x = (float)c - W / 2.0f, y = H / 2.0f - (float)r;
                const float zb = zimg[pp];
                float n[3];
                if (a.normals) {
                    n[0] = a.normals[((size_t)b * 3 + 0) * P + pp];
                    n[1] = a.normals[((size_t)b * 3 + 1) * P + pp];
                    n[2] = a.normals[((size_t)b * 3 + 2) * P + pp];
                } else {
                    n[0] = npre[0];
                    n[1] = npre[1];
                    n[2] = npre[2];
                }
                float nn = __builtin_sqrtf(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]);
                nn = nn > 1e-12f ? nn : 1e-12f;
                const float inv_nn = 1.0f / nn;
                const float n0 = n[0] * inv_nn, n1 = n[1] * inv_nn, n2 = n[2] * inv_nn;
                const float amb = a.ambient[bl];
                const float lx = Cxf - x, ly = Cyf - y, lz = Czf - zb;
                float ln = __builtin_sqrtf(lx * lx + ly * ly + lz * lz);
                ln = ln > 1e-12f ? ln : 1e-12f;
                const float inv_ln = 1.0f / ln;
                const float u0 = lx * inv_ln, u1 = ly * inv_ln, u2 = lz * inv_ln;
                float dot = n0 * u0 + n1 * u1 + n2 * u2;
                if (fabsf(dot) < 1e-4f)  // at the kink of max(n.l, 0) the forward's own arithmetic decides the side (lambert_dot)
                    dot = lambert_dot(x, y, zb, n[0], n[1], n[2], Cxf, Cyf, Czf);
                const float full = amb + a.intensity * (dot > 0.0f ? dot : 0.0f);
                const float e = expf(-a.min_dist[o]);  // T8:517
                const float ope = 1.0f + e;
                const float inv_ope = 1.0f / ope, inv_ope2 = inv_ope * inv_ope;
                const float w = 1.0f - 4.0f * e * inv_ope2;
                const float fin = w * full + (1.0f - w) * amb;
                const float al0 = a.albedo[((size_t)b * 3 + 0) * P + pp];
                const float al1 = a.albedo[((size_t)b * 3 + 1) * P + pp];
                const float al2 = a.albedo[((size_t)b * 3 + 2) * P + pp];
                ga0 += gr0 * fin;  // rendered = albedo * final  (T8:520-522)
                ga1 += gr1 * fin;
                ga2 += gr2 * fin;
                const float dfin = (gr0 * al0 + gr1 * al1 + gr2 * al2) + gfin;
                const float dw = dfin * (full - amb) + gw;  // final = w*full + (1-w)*amb (T8:518)
                const float dfull = dfin * w + gfull;
                red[3] = (double)(dfin * (1.0f - w) + dfull);
                gmd = dw * (4.0f * e * (1.0f - e)) * (inv_ope2 * inv_ope);  // dw/dd = 4e(1-e)/(1+e)^3 (T8:517)
                const float ddot = (dot > 0.0f) ? dfull * a.intensity : 0.0f;  // full = amb + I*max(dot,0) (T8:366)
                const float dn0 = ddot * u0, dn1 = ddot * u1, dn2 = ddot * u2;
                const float du0 = ddot * n0, du1 = ddot * n1, du2 = ddot * n2;
                const float nd = n0 * dn0 + n1 * dn1 + n2 * dn2;  // n_hat = n/|n|
                h0 += (dn0 - n0 * nd) * inv_nn;
                h1 += (dn1 - n1 * nd) * inv_nn;
                h2 += (dn2 - n2 * nd) * inv_nn;
                const float ud = u0 * du0 + u1 * du1 + u2 * du2;  // l_hat = l/|l|, l = C - P
                const float dl2 = (du2 - u2 * ud) * inv_ln;
                red[0] = (double)((du0 - u0 * ud) * inv_ln);
                red[1] = (double)((du1 - u1 * ud) * inv_ln);
                red[2] = (double)dl2;
                gzb -= (double)dl2;
            }
        }
        if (live) {  // ---- (2) ray-march backward through the argmin sample ----
            const int k = a.argmin[(size_t)bl * P + pp];
            if (k >= 0 && k < a.N && gmd != 0.0f) {
                double gC[3];
                shadow_bwd_pixel(zimg, gz, a.t_table, H, W, r, c, Cxf, Cyf, Czf, k, gmd, gC, &gzb, nullptr);
                red[0] += gC[0];
                red[1] += gC[1];
                red[2] += gC[2];
            }
        }
        block_reduce_atomic4(red, a.grad_light_pt + 3 * (size_t)bl, a.grad_ambient + bl);  // ---- (3) ----
    }
    if (live) {  // ---- (4) stencil backward, once per pixel ----
        double g0 = (double)h0, g1 = (double)h1, g2 = (double)h2;
        if (a.g_normals_out) {
            g0 += a.g_normals_out[((size_t)b * 3 + 0) * P + p];
            g1 += a.g_normals_out[((size_t)b * 3 + 1) * P + p];
            g2 += a.g_normals_out[((size_t)b * 3 + 2) * P + p];
        }
        if (g0 != 0.0 || g1 != 0.0 || g2 != 0.0)
            normals_bwd_pixel(a.nrm, zimg, gz, r, c, g0, g1, g2);
        a.grad_albedo[((size_t)b * 3 + 0) * P + p] = ga0;
        a.grad_albedo[((size_t)b * 3 + 1) * P + p] = ga1;
        a.grad_albedo[((size_t)b * 3 + 2) * P + p] = ga2;
        if (gzb != 0.0)
            atomicAdd(gz + p, (float)gzb);
    }
}

}  // namespace gcfr

using namespace gcfr;

static inline int launch_status() { return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH; }

extern "C" int gcfr_shadow_bwd(const float *grad_min_dist, const float *depth, const float *light_pt,
                               const int32_t *argmin, int32_t B, int32_t L, int32_t H, int32_t W,
                               int32_t N, const double *t_table, float *grad_depth,
                               double *grad_light_pt, void *stream)
{
    if (!grad_min_dist || !depth || !light_pt || !argmin || !t_table || !grad_depth || !grad_light_pt)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || N <= 0 || H < 2 || W < 2 || H > 4096 || W > 4096 || (H & 1) || (W & 1) ||
        (long long)B * L > 65535)
        return GCFR_ERR_INVALID_ARGUMENT;
    ShadowBwdArgs a{grad_min_dist, depth, light_pt, argmin, t_table, grad_depth, grad_light_pt, L, H, W, N};
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL(shadow_bwd_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(B * L)), dim3(256),
                       0, (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int gcfr_shade_bwd(const float *normals, const float *depth, const float *albedo,
                              const float *light_pt, const float *ambient, const float *min_dist,
                              int32_t B, int32_t L, int32_t H, int32_t W, float intensity,
                              const float *g_shadow_w, const float *g_full, const float *g_final,
                              const float *g_rendered, float *grad_normals, float *grad_albedo,
                              float *grad_depth, double *grad_light_pt, double *grad_ambient,
                              float *grad_min_dist, void *stream)
{
    if (!normals || !depth || !albedo || !light_pt || !ambient || !min_dist || !grad_normals ||
        !grad_albedo || !grad_depth || !grad_light_pt || !grad_ambient || !grad_min_dist)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || H <= 0 || W <= 0 || B > 65535)
        return GCFR_ERR_INVALID_ARGUMENT;
    ShadeBwdArgs a{};
    a.normals = normals;
    a.depth = depth;
    a.albedo = albedo;
    a.light_pt = light_pt;
    a.ambient = ambient;
    a.min_dist = min_dist;
    a.g_w = g_shadow_w;
    a.g_full = g_full;
    a.g_final = g_final;
    a.g_rendered = g_rendered;
    a.grad_normals = grad_normals;
    a.grad_albedo = grad_albedo;
    a.grad_depth = grad_depth;
    a.grad_light_pt = grad_light_pt;
    a.grad_ambient = grad_ambient;
    a.grad_min_dist = grad_min_dist;
    a.L = L;
    a.H = H;
    a.W = W;
    a.intensity = intensity;
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL(shade_bwd_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, a);
    return launch_status();
}

extern "C" int gcfr_render_bwd(const float *depth, const float *albedo, const float *light_pt,
                               const float *ambient, const float *min_dist, const int32_t *argmin,
                               const float *normals_fwd, int32_t B,
                               int32_t L, int32_t H, int32_t W, int32_t N, const double *t_table, double fx,
                               double fy, double cx, double cy, float z_offset, int32_t negate_y,
                               float intensity, const float *g_shadow_w, const float *g_full,
                               const float *g_final, const float *g_rendered, const float *g_normals_out,
                               float *grad_albedo, float *grad_depth, double *grad_light_pt,
                               double *grad_ambient, void *stream)
{
    if (!depth || !albedo || !light_pt || !ambient || !min_dist || !argmin || !t_table || !grad_albedo ||
        !grad_depth || !grad_light_pt || !grad_ambient)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || N <= 0 || H < 2 || W < 2 || H > 4096 || W > 4096 || (H & 1) || (W & 1) || B > 65535 ||
        fx == 0.0 || fy == 0.0)
        return GCFR_ERR_INVALID_ARGUMENT;
    ShadeBwdArgs a{};
    a.depth = depth;
    a.albedo = albedo;
    a.light_pt = light_pt;
    a.ambient = ambient;
    a.min_dist = min_dist;
    a.g_w = g_shadow_w;
    a.g_full = g_full;
    a.g_final = g_final;
    a.g_rendered = g_rendered;
    a.grad_albedo = grad_albedo;
    a.grad_depth = grad_depth;
    a.grad_light_pt = grad_light_pt;
    a.grad_ambient = grad_ambient;
    a.L = L;
    a.H = H;
    a.W = W;
    a.intensity = intensity;
    a.argmin = argmin;
    a.normals = normals_fwd;  // the unit normals the forward wrote, or NULL: recomputed from the depth stencil
    a.t_table = t_table;
    a.N = N;
    a.nrm.depth = depth;
    a.nrm.H = H;
    a.nrm.W = W;
    set_focal(a.nrm, fx, fy);
    a.nrm.cx = cx;
    a.nrm.cy = cy;
    a.nrm.z_offset = z_offset;
    a.nrm.negate_y = negate_y;
    a.g_normals_out = g_normals_out;
    const size_t P = (size_t)H * W;
    if (L == 1)  // the training shape: staged kernel, higher occupancy (same device functions, same numbers)
    {
        // workgroups per image: enough to fill the chip about twice (256 CUs x 4 resident workgroups), at most one per tile
        const int n_tiles = ((W + kBwdTileW - 1) / kBwdTileW) * ((H + kBwdTileH - 1) / kBwdTileH);
        int per_image = (2048 + B - 1) / B;
        per_image = per_image < 1 ? 1 : (per_image > n_tiles ? n_tiles : per_image);
        hipLaunchKernelGGL(render_bwd_single_light_kernel, dim3((unsigned)per_image, (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, a);
    }
    else
        hipLaunchKernelGGL(render_bwd_multi_light_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0,
                           (hipStream_t)stream, a);
    return launch_status();
}

#ifdef GCFR_BWD_COUNT
// counting build only: device buffer of 3 * 4 u64 (zeroed by the caller) the fused backward adds its request tallies to; NULL = off
extern "C" int gcfr_debug_set_bwd_count(unsigned long long *device_buffer)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(gcfr::g_bwd_count), &device_buffer, sizeof(device_buffer)) == hipSuccess ? 0 : -2;
}
#endif

#ifdef GCFR_BWD_TRACE
extern "C" int gcfr_debug_set_bwd_trace(unsigned long long *device_buffer)
{
    return hipMemcpyToSymbol(HIP_SYMBOL(gcfr::g_bwd_trace), &device_buffer, sizeof(device_buffer)) == hipSuccess ? 0 : -2;
}
#endif

extern "C" int gcfr_light_prep_bwd(const float *light_raw, int32_t n, int32_t clamp_z, float clamp_min,
                                   float light_distance, const float *grad_unit,
                                   const double *grad_light_pt, float *grad_light_raw, void *stream)
{
    if (!light_raw || !grad_light_raw || n <= 0 || (!grad_unit && !grad_light_pt))
        return GCFR_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(light_prep_bwd_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       light_raw, n, clamp_z, clamp_min, light_distance, grad_unit, grad_light_pt,
                       grad_light_raw);
    return launch_status();
}
