// Ray-marched minimum point-to-line distance (the hot kernel) + light preparation, gfx950.
//
// Replaces train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:357-363 and :371-515.  The reference
// materialises (N,2,H,W) f64 sample grids per image (2.8 GB of temporaries per 256x256 face); here
// one lane owns one pixel and walks its N samples in registers:
//   - one wavefront = one TILE_H x TILE_W pixel tile (64 lanes); neighbouring lanes march
//     neighbouring, nearly parallel rays, so every gather instruction of the wave touches a
//     footprint about the size of the tile -> a handful of 128-B lines served by the CU's L1;
//   - the per-image working set (H*W*4 B depth + H*W B mask = 320 KB at 256x256) lives in L2, so
//     HBM traffic is compulsory only; the kernel is VALU / vector-memory-issue bound, not HBM bound;
//   - the sample table is wave-uniform: it is read with scalar loads (SMEM), costing no VALU;
//   - gathers use raw buffer loads (32-bit offsets off an SGPR descriptor: no 64-bit address VALU,
//     hardware range check instead of per-sample clamps).
// No MFMA: there is no dense contraction anywhere on this path.
#include "gcfr_device.hpp"

#include "../../include/gcfr.h"

namespace gcfr {

// ----------------------------------------------------------------------------------------------
// light preparation, T8:357-363 / S1:332-336
// ----------------------------------------------------------------------------------------------
__global__ void light_prep_kernel(const float *__restrict__ light_raw, int n, int clamp_z,
                                  float clamp_min, float light_distance, float *__restrict__ unit_out,
                                  float *__restrict__ light_pt_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    const float a = light_raw[3 * i + 0], b = light_raw[3 * i + 1];
    float c = light_raw[3 * i + 2];
    if (clamp_z)
        c = (c > clamp_min) ? c : clamp_min;  // torch.maximum(l_z, 0)  T8:358
    const float nrm = norm3_torch(a, b, c);   // F.normalize          T8:360
    const float d = (nrm > 1e-12f) ? nrm : 1e-12f;
    const float ux = a / d, uy = b / d, uz = c / d;
    unit_out[3 * i + 0] = ux;
    unit_out[3 * i + 1] = uy;
    unit_out[3 * i + 2] = uz;
    light_pt_out[3 * i + 0] = light_distance * ux;  // T8:362
    light_pt_out[3 * i + 1] = light_distance * uy;
    light_pt_out[3 * i + 2] = light_distance * uz;
}

// ----------------------------------------------------------------------------------------------
// shadow march
// ----------------------------------------------------------------------------------------------
struct ShadowArgs {
    const float *depth;       // (B,H,W)
    const uint8_t *mask;      // (MB,H,W)
    const float *light_pt;    // (B,L,3)
    const double *t_table;    // (N)
    float *min_dist;          // (B,L,H,W)
    int32_t *argmin;          // (B,L,H,W) or null
    int32_t mask_batch, L, H, W, N;
    int32_t tiles_x, tiles_per_image;
    float bonus, bx_lo, bx_hi, by_lo, by_hi;
};

// One sample of one ray: returns S = |BA x BC|^2 + 1e-4 (f32) and whether the sample is masked.
// Position pipeline in f64 exactly as T8:472-502; distance in f32 as T8:504-509.
struct RayConst {
    float x, y, zb;        // pixel B (T8:503)
    float dx, dy;          // end - start (T8:467)
    float BCx, BCy, BCz;   // light - pixel (T8:507)
};

template <int TILE_W>
__global__ __launch_bounds__(256) void shadow_fwd_kernel(ShadowArgs a)
{
    constexpr int TILE_H = 64 / TILE_W;
    constexpr int WAVES = 4;  // 256 threads; the four tiles of a block are horizontal neighbours
    const int H = a.H, W = a.W, N = a.N;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;

    // blockIdx.x enumerates (image*light, tile-quad) pairs.
    const int quads_x = (a.tiles_x + WAVES - 1) / WAVES;
    const int quads_per_image = quads_x * ((H + TILE_H - 1) / TILE_H);
    const int bl = blockIdx.x / quads_per_image;
    const int q = blockIdx.x - bl * quads_per_image;
    const int qy = q / quads_x, qx = q - qy * quads_x;
    const int b = bl / a.L;

    int r = qy * TILE_H + lane / TILE_W;
    int c = (qx * WAVES + wave) * TILE_W + (lane % TILE_W);
    const bool valid = (r < H) && (c < W);
    r = valid ? r : H - 1;  // keep the wave convergent; out-of-image lanes redo a border pixel
    c = valid ? c : W - 1;

    const size_t P = (size_t)H * W;
    const float *zimg = a.depth + (size_t)b * P;
    const __amdgpu_buffer_rsrc_t zr = make_rsrc(zimg, (int)(P * 4));
    const __amdgpu_buffer_rsrc_t mr =
        make_rsrc(a.mask + (size_t)(a.mask_batch == 1 ? 0 : b) * P, (int)P);

    const float Cx = a.light_pt[3 * bl + 0], Cy = a.light_pt[3 * bl + 1], Cz = a.light_pt[3 * bl + 2];
    const Box box = image_box(H, W);
    const LightCase lc = classify_light(Cx, Cy, box);

    const float halfWf = W / 2.0f, halfHf = H / 2.0f;
    const double halfW = W / 2.0, halfH = H / 2.0;

    RayConst rc;
    rc.x = (float)c - halfWf;  // T8:52
    rc.y = halfHf - (float)r;  // T8:53
    rc.zb = zimg[(size_t)r * W + c];
    float Ex, Ey;
    end_point(rc.x, rc.y, Cx, Cy, box, lc, Ex, Ey);
    rc.dx = Ex - rc.x;
    rc.dy = Ey - rc.y;
    rc.BCx = Cx - rc.x;
    rc.BCy = Cy - rc.y;
    rc.BCz = Cz - rc.zb;
    const bool finite_ray = (rc.dx - rc.dx == 0.0f) && (rc.dy - rc.dy == 0.0f);
    const double x64 = rc.x, y64 = rc.y;
    const double dx64 = finite_ray ? (double)rc.dx : 0.0, dy64 = finite_ray ? (double)rc.dy : 0.0;

    // sqrt and the division by the per-pixel constant |BC| are monotone, so
    // min_k sqrt(S_k)/den == sqrt(min_k S_k)/den bit for bit: track the minimum of S over the
    // unmasked samples and finish once per pixel.
    float bestS = __builtin_inff();
    int besti = -1;
    bool any_masked = false;

    for (int k = 0; k < N; ++k) {
        const double t = a.t_table[k];  // wave-uniform -> s_load
        const double sx = x64 + t * dx64;  // T8:472 / 480 (f64, mul and add rounded separately)
        const double sy = y64 + t * dy64;
        // rounded cell -> mask lookup (T8:472-477, 510)
        const int col_r = (int)(__builtin_rint(sx) + halfW);
        const int row_r = (int)(halfH - __builtin_rint(sy));
        // unrounded position (T8:480-487)
        const double ux = (sx + halfW) - 0.0001;
        const double uy = (halfH - sy) - 0.0001;
        const double fxd = __builtin_floor(ux), gxd = __builtin_ceil(ux);
        const double fyd = __builtin_floor(uy), gyd = __builtin_ceil(uy);
        int fx = (int)fxd, gx = (int)gxd, fy = (int)fyd, gy = (int)gyd;
        const double wx0 = gxd - ux, wx1 = ux - fxd;  // T8:492-494 weights
        const double wy0 = gyd - uy, wy1 = uy - fyd;
        fx += (fx >> 31) & W;  // index -1 wraps to W-1 / H-1 (T8:488-491)
        fy += (fy >> 31) & H;
        const int rowf = fy * W, rowg = gy * W;
        const double zUL = buf_load_f32(zr, (rowf + fx) << 2);
        const double zUR = buf_load_f32(zr, (rowf + gx) << 2);
        const double zLL = buf_load_f32(zr, (rowg + fx) << 2);
        const double zLR = buf_load_f32(zr, (rowg + gx) << 2);
        const uint32_t mk = buf_load_u8(mr, row_r * W + col_r);
        const double up = zUL * wx0 + zUR * wx1;
        const double low = zLL * wx0 + zLR * wx1;
        const double zA = up * wy0 + low * wy1;
        // point A (T8:497-502) and the distance numerator (T8:504-509) in f32
        const float Ax = (float)(ux - halfW), Ay = (float)(halfH - uy), Az = (float)zA;
        const float BAx = Ax - rc.x, BAy = Ay - rc.y, BAz = Az - rc.zb;
        const float Xx = __builtin_fmaf(BAy, rc.BCz, -(BAz * rc.BCy));  // torch.cross uses fma
        const float Xy = __builtin_fmaf(BAz, rc.BCx, -(BAx * rc.BCz));
        const float Xz = __builtin_fmaf(BAx, rc.BCy, -(BAy * rc.BCx));
        const float S = ((Xx * Xx + Xy * Xy) + Xz * Xz) + kEps4;
        const bool masked = (mk == 0);
        any_masked |= masked;
        const bool take = !masked && (S < bestS);  // strict: first minimum wins (T8:514)
        bestS = take ? S : bestS;
        besti = take ? k : besti;
    }

    const float den = __builtin_sqrtf(((rc.BCx * rc.BCx + rc.BCy * rc.BCy) + rc.BCz * rc.BCz) + kEps4);
    float d = __builtin_sqrtf(bestS) / den;  // +inf when every sample was masked
    if (any_masked && !(d < kMaskedDistance)) {  // T8:512: masked samples count as 1e6
        d = kMaskedDistance;
        besti = -1;  // no gradient flows through a masked minimum
    }
    if (!finite_ray)
        d = __builtin_nanf("");  // the reference raises here (NaN index); see DESIGN.md
    const bool inside = (Cx >= a.bx_lo) && (Cx <= a.bx_hi) && (Cy >= a.by_lo) && (Cy <= a.by_hi);
    if (inside)
        d = d + a.bonus;  // S1:495-496

    if (valid) {
        const size_t o = (size_t)bl * P + (size_t)r * W + c;
        a.min_dist[o] = d;
        if (a.argmin)
            a.argmin[o] = besti;
    }
}

}  // namespace gcfr

// ----------------------------------------------------------------------------------------------
// C ABI
// ----------------------------------------------------------------------------------------------
using namespace gcfr;

static inline int launch_status()
{
    return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH;
}

extern "C" const char *gcfr_version(void) { return "gcfr-hip 0.1.0 gfx950"; }

extern "C" int gcfr_sample_table(double t0, double dt, int32_t n, double *out_host)
{
    if (!out_host || n <= 0)
        return GCFR_ERR_INVALID_ARGUMENT;
    const volatile double second = t0 + dt;  // numpy: delta = (start + step) - start
    const double delta = second - t0;
    for (int k = 0; k < n; ++k)
        out_host[k] = t0 + (double)k * delta;
    return GCFR_OK;
}

extern "C" int gcfr_light_prep(const float *light_raw, int32_t n, int32_t clamp_z, float clamp_min,
                               float light_distance, float *unit_out, float *light_pt_out,
                               void *stream)
{
    if (!light_raw || !unit_out || !light_pt_out || n <= 0)
        return GCFR_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(light_prep_kernel, dim3((n + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       light_raw, n, clamp_z, clamp_min, light_distance, unit_out, light_pt_out);
    return launch_status();
}

extern "C" int gcfr_shadow_fwd(const float *depth, const uint8_t *mask_u8, int32_t mask_batch,
                               const float *light_pt, int32_t B, int32_t L, int32_t H, int32_t W,
                               int32_t N, const double *t_table, float bonus,
                               const float *bonus_box, float *min_dist, int32_t *argmin,
                               void *stream)
{
    if (!depth || !mask_u8 || !light_pt || !t_table || !min_dist)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || N <= 0 || N > 4096 || H < 2 || W < 2 || H > 4096 || W > 4096 ||
        (H & 1) || (W & 1) || (mask_batch != 1 && mask_batch != B))
        return GCFR_ERR_INVALID_ARGUMENT;
    if (bonus != 0.0f && !bonus_box)
        return GCFR_ERR_INVALID_ARGUMENT;

    constexpr int TILE_W = 16, TILE_H = 64 / TILE_W, WAVES = 4;
    ShadowArgs a;
    a.depth = depth;
    a.mask = mask_u8;
    a.light_pt = light_pt;
    a.t_table = t_table;
    a.min_dist = min_dist;
    a.argmin = argmin;
    a.mask_batch = mask_batch;
    a.L = L;
    a.H = H;
    a.W = W;
    a.N = N;
    a.tiles_x = (W + TILE_W - 1) / TILE_W;
    const int quads_x = (a.tiles_x + WAVES - 1) / WAVES;
    const int quads_per_image = quads_x * ((H + TILE_H - 1) / TILE_H);
    a.tiles_per_image = quads_per_image;
    a.bonus = bonus;
    if (bonus_box) {
        a.bx_lo = bonus_box[0];
        a.bx_hi = bonus_box[1];
        a.by_lo = bonus_box[2];
        a.by_hi = bonus_box[3];
    } else {
        a.bx_lo = a.by_lo = 0.0f;
        a.bx_hi = a.by_hi = -1.0f;
    }
    const long long blocks = (long long)B * L * quads_per_image;
    if (blocks > 0x7fffffffLL)
        return GCFR_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(shadow_fwd_kernel<TILE_W>, dim3((unsigned)blocks), dim3(256), 0,
                       (hipStream_t)stream, a);
    return launch_status();
}
