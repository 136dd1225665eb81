// Soft-shadow transfer + Lambertian shading + albedo compositing, gfx950.
//
// Replaces train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:364-369 and :517-522.  Pure streaming:
// every operand is read once and every output written once with fully coalesced plane accesses
// (one lane per pixel, consecutive lanes = consecutive columns), so this kernel sits on the HBM
// roofline: 36 B read (normals 12, albedo 12, depth 4, min_dist 4 [+light/ambient in SGPRs]) and
// 24 B written (w, full, final, rendered x3) per pixel.
//
// dtypes: the reference's normals are f64 only because its camera matrix is f64 (torch promotion);
// here normals arrive as f32 and the arithmetic is f32.  The resulting differences are ~1e-7,
// four orders of magnitude inside the 1e-3 RGB gate.
#include "gcfr_device.hpp"

#include "../../include/gcfr.h"

namespace gcfr {

struct ShadeArgs {
    const float *normals;   // (B,3,H,W)
    const float *depth;     // (B,H,W)
    const float *albedo;    // (B,3,H,W)
    const float *light_pt;  // (B,L,3)
    const float *ambient;   // (B,L)
    const float *min_dist;  // (B,L,H,W)
    float *shadow_w, *full, *final_shading, *rendered;
    int32_t L, H, W;
    float intensity;
};

__global__ __launch_bounds__(256) void shade_fwd_kernel(ShadeArgs a)
{
    const int W = a.W, H = a.H;
    const size_t P = (size_t)H * W;
    const int bl = blockIdx.y;
    const int b = bl / a.L;
    const float Cx = a.light_pt[3 * bl + 0], Cy = a.light_pt[3 * bl + 1], Cz = a.light_pt[3 * bl + 2];
    const float amb = a.ambient[bl];
    const float halfWf = W / 2.0f, halfHf = H / 2.0f;

    for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < P;
         p += (size_t)gridDim.x * blockDim.x) {
        const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
        const float x = (float)c - halfWf, y = halfHf - (float)r;
        const float zb = a.depth[(size_t)b * P + p];
        const Shaded sh = shade_pixel(x, y, zb, a.normals[((size_t)b * 3 + 0) * P + p],
                                      a.normals[((size_t)b * 3 + 1) * P + p],
                                      a.normals[((size_t)b * 3 + 2) * P + p], Cx, Cy, Cz, amb, a.intensity,
                                      a.min_dist[(size_t)bl * P + p]);
        const float w = sh.w, full = sh.full, fin = sh.fin;
        const size_t o = (size_t)bl * P + p;
        if (a.shadow_w)
            a.shadow_w[o] = w;
        if (a.full)
            a.full[o] = full;
        if (a.final_shading)
            a.final_shading[o] = fin;
#pragma unroll
        for (int ch = 0; ch < 3; ++ch)  // T8:519-522
            a.rendered[((size_t)bl * 3 + ch) * P + p] = a.albedo[((size_t)b * 3 + ch) * P + p] * fin;
    }
}

}  // namespace gcfr

using namespace gcfr;

extern "C" int gcfr_shade_fwd(const float *normals, const float *depth, const float *albedo,
                              const float *light_pt, const float *ambient, const float *min_dist,
                              int32_t B, int32_t L, int32_t H, int32_t W, float intensity,
                              float *shadow_w, float *full, float *final_shading, float *rendered,
                              void *stream)
{
    if (!normals || !depth || !albedo || !light_pt || !ambient || !min_dist || !rendered)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || H <= 0 || W <= 0 || (long long)B * L > 65535)
        return GCFR_ERR_INVALID_ARGUMENT;
    ShadeArgs a{normals, depth,  albedo, light_pt, ambient,      min_dist, shadow_w,
                full,    final_shading, rendered, L, H, W, intensity};
    const size_t P = (size_t)H * W;
    unsigned gx = (unsigned)((P + 255) / 256);
    if (gx > 1024)
        gx = 1024;
    hipLaunchKernelGGL(shade_fwd_kernel, dim3(gx, (unsigned)(B * L)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH;
}
