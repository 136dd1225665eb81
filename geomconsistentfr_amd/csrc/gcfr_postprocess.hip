// What the reference's inference scripts do with the render block's outputs before cv2.imwrite, gfx950:
// paste the relit face into the input photograph, scale the diagnostic maps, quantise to u8 (S1:601-620,
// S8:583-608, SLT:560-579), and the MATLAB border fix (fix_border_artifacts_CVPR2022.m:1-18).
//
// Byte / elementwise work, HBM-bound by construction: one lane per pixel, every operand read once from coalesced
// planes, every u8 image written once -- so a relit batch stays on the device until the PNG encoder wants bytes
// (the reference bounces every tensor through host numpy first).  No MFMA, no LDS (the 7x7 box sum and the 3x3
// median read neighbours through L1: 49 + 27 cached byte loads on the ~3 % of pixels that form the mask's border).
//
// Arithmetic follows the scripts' numpy expressions op for op with the dtypes the REFERENCE holds, because a half-way
// case decides a byte (pinned to the reference's own main(): tests/golden/slt_main_*.npz, oracle/make_golden_slt_main.py):
//   255.0*rendered / albedo   f32 * python float -> f32 (numpy keeps the array's dtype)
//   (...)*mask_3_channels     f32 * f64 -> f64   (np.zeros((H,W,3)) is f64, S1:602)
//   final_shading, normals    f64 in the reference (promotion from the f64 camera matrix): the device holds them as f32
//                             and they are widened BEFORE the arithmetic, so 255.0*x*mask runs in f64 as in the scripts
//   the mask                  mask_f32 = 0: f64, a numpy f64 array / 255.0 (S1:580, S8:569-578);
//                             mask_f32 = 1: f32, a torch uint8 tensor / 255.0 (SLT:540) -- the 3-channel mask is still an
//                             f64 array holding those f32 values, but the single-channel products of f32 maps (shadow
//                             mask SLT:575, depth map SLT:577) then stay f32
//   training_images*255.0     f64 (imread / 255.0 is f64, S1:513-516); here the image arrives as f32 and is widened
//   cv2.imwrite(float image)  saturate_cast<uchar>(cvRound(v)): round half to even, clip to [0, 255]
// Channel order: the scripts flip to BGR only because cv2 writes BGR; the bytes produced here are RGB, HWC -- what
// ends up in the PNG.
#include "gcfr_device.hpp"

#include "../../include/gcfr.h"

namespace gcfr {

__device__ inline uint8_t quantise_u8(double v)
{
    const double r = __builtin_rint(v);  // cvRound: round half to even
    return (uint8_t)(r < 0.0 ? 0.0 : (r > 255.0 ? 255.0 : r));  // NaN -> 0 (saturate_cast of INT_MIN)
}

struct ImagesArgs {
    const float *input_hwc;   // (B,H,W,3) f32 in [0,1]
    const float *rendered;    // (B,L,3,H,W): L relit images per photograph
    const float *albedo;      // (B,3,H,W) or null
    const float *depth;       // (B,H,W) or null
    const float *depth_range; // device {min, max} of -depth over the whole batch (S8:589-590), with depth
    const float *shadow_w;    // (B,L,H,W) or null
    const float *shading;     // (B,L,H,W) or null
    const float *normals;     // (B,3,H,W) or null
    const uint8_t *mask;      // (MB,H,W) u8 skin mask as stored on disk; mask/255.0 is an f64 division (S1:580)
    uint8_t *out_rendered;    // (B,L,H,W,3)
    uint8_t *out_shadow, *out_albedo, *out_depth, *out_shading, *out_normals;  // shadow / shading (B,L,H,W); the per-photograph
                                                                               // maps (B,H,W) / (B,H,W,3); or null
    int32_t mask_batch, L, H, W, mask_f32;
};

__global__ __launch_bounds__(256) void inference_images_kernel(ImagesArgs a)
{
    const size_t P = (size_t)a.H * a.W;
    const int bl = blockIdx.y;             // (photograph, light) pair; the per-photograph maps are written by light 0's threads
    const int b = bl / a.L;
    const bool first = bl - b * a.L == 0;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P)
        return;
    const uint8_t mk = a.mask[(size_t)(a.mask_batch == 1 ? 0 : b) * P + p];
    const float mf = (float)mk / 255.0f;                                   // SLT:540 (torch: u8 -> f32, IEEE division)
    const double m = a.mask_f32 ? (double)mf : (double)mk / 255.0;         // value of mask_3_channels (an f64 array)
    const size_t hwc = ((size_t)b * P + p) * 3, hwc_l = ((size_t)bl * P + p) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        // input_image = training_images*255.0; rendered_image = 255.0*rendered*mask; input[mask > 0] = rendered[mask > 0]
        const double keep = (double)a.input_hwc[hwc + ch] * 255.0;
        const double paste = (double)(255.0f * a.rendered[((size_t)bl * 3 + ch) * P + p]) * m;
        a.out_rendered[hwc_l + ch] = quantise_u8(m > 0.0 ? paste : keep);
        if (a.out_albedo && first)   // 255.0*albedo*mask3                                   S8:605
            a.out_albedo[hwc + ch] = quantise_u8((double)(255.0f * a.albedo[((size_t)b * 3 + ch) * P + p]) * m);
        if (a.out_normals && first)  // (255.0*(n + 1.0)/2.0)*mask3, n f64 in the reference  S8:594, 608
            a.out_normals[hwc + ch] = quantise_u8(((255.0 * ((double)a.normals[((size_t)b * 3 + ch) * P + p] + 1.0)) / 2.0) * m);
    }
    const size_t o = (size_t)b * P + p, o_l = (size_t)bl * P + p;
    if (a.out_shadow) {  // 255.0*shadow_mask_weights*mask (single-channel mask in ITS dtype)   S8:604 / SLT:575
        const float s255 = 255.0f * a.shadow_w[o_l];
        a.out_shadow[o_l] = quantise_u8(a.mask_f32 ? (double)(s255 * mf) : (double)s255 * m);
    }
    if (a.out_shading)  // 255.0*final_shading*mask, final_shading f64 in the reference          S8:607
        a.out_shading[o_l] = quantise_u8((255.0 * (double)a.shading[o_l]) * m);
    if (a.out_depth && first) {  // depth = -depth; (depth - amin)/(amax - amin) in f32; 255.0*depth*mask   S8:588-590, 606
        const float lo = a.depth_range[0], hi = a.depth_range[1];
        const float d = ((-a.depth[o]) - lo) / (hi - lo);
        const float d255 = 255.0f * d;
        a.out_depth[o] = quantise_u8(a.mask_f32 ? (double)(d255 * mf) : (double)d255 * m);
    }
}

// fix_border_artifacts_CVPR2022.m: face_mask = imread(mask)/255.0 is a UINT8 division (rounds to nearest: 64 -> 0,
// 128 -> 1, 255 -> 1); convolved = imfilter(double(face_mask), ones(7,7)) with zero padding; border = 0 < convolved < 30;
// border pixels take medfilt2's 3x3 median (zero padding) of their channel.
__device__ inline uint8_t median9(uint8_t (&v)[9])
{
    // partial selection network: after it v[4] is the median of the nine
#define GCFR_SWAP(i, j)                         \
    {                                           \
        const uint8_t lo = min(v[i], v[j]);     \
        const uint8_t hi = max(v[i], v[j]);     \
        v[i] = lo;                              \
        v[j] = hi;                              \
    }
    GCFR_SWAP(1, 2) GCFR_SWAP(4, 5) GCFR_SWAP(7, 8) GCFR_SWAP(0, 1) GCFR_SWAP(3, 4) GCFR_SWAP(6, 7)
    GCFR_SWAP(1, 2) GCFR_SWAP(4, 5) GCFR_SWAP(7, 8) GCFR_SWAP(0, 3) GCFR_SWAP(5, 8) GCFR_SWAP(4, 7)
    GCFR_SWAP(3, 6) GCFR_SWAP(1, 4) GCFR_SWAP(2, 5) GCFR_SWAP(4, 7) GCFR_SWAP(4, 2) GCFR_SWAP(6, 4)
    GCFR_SWAP(4, 2)
#undef GCFR_SWAP
    return v[4];
}

__global__ __launch_bounds__(256) void fix_border_kernel(const uint8_t *__restrict__ img, const uint8_t *__restrict__ mask,
                                                         int mask_batch, int H, int W, uint8_t *__restrict__ out)
{
    const size_t P = (size_t)H * W;
    const int b = blockIdx.y;
    const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P)
        return;
    const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
    const uint8_t *mk = mask + (size_t)(mask_batch == 1 ? 0 : b) * P;
    const uint8_t *im = img + (size_t)b * P * 3;
    int sum = 0;
    for (int dr = -3; dr <= 3; ++dr)
        for (int dc = -3; dc <= 3; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if (rr >= 0 && rr < H && cc >= 0 && cc < W)
                sum += mk[(size_t)rr * W + cc] >= 128 ? 1 : 0;  // uint8(x)/255.0 rounds: x >= 127.5 -> 1
        }
    const bool border = (sum > 0) && (sum < 30);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        uint8_t v = im[p * 3 + ch];
        if (border) {  // (about 3 % of a face image)
            uint8_t n[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) {
                const int rr = r + q / 3 - 1, cc = c + q % 3 - 1;
                n[q] = (rr >= 0 && rr < H && cc >= 0 && cc < W) ? im[((size_t)rr * W + cc) * 3 + ch] : (uint8_t)0;
            }
            v = median9(n);
        }
        out[((size_t)b * P + p) * 3 + ch] = v;
    }
}

}  // namespace gcfr

using namespace gcfr;

extern "C" int gcfr_inference_images_u8(const float *input_hwc, const float *rendered, const float *albedo, const float *depth,
                                        const float *depth_range, const float *shadow_w, const float *final_shading,
                                        const float *normals, const uint8_t *mask, int32_t mask_batch, int32_t B, int32_t L,
                                        int32_t H, int32_t W, uint8_t *out_rendered, uint8_t *out_shadow, uint8_t *out_albedo,
                                        uint8_t *out_depth, uint8_t *out_shading, uint8_t *out_normals, int32_t mask_f32,
                                        void *stream)
{
    if (!input_hwc || !rendered || !mask || !out_rendered || B <= 0 || L <= 0 || H <= 0 || W <= 0 || (int64_t)B * L > 65535 ||
        (mask_batch != 1 && mask_batch != B) || (mask_f32 != 0 && mask_f32 != 1))
        return GCFR_ERR_INVALID_ARGUMENT;
    if ((out_shadow && !shadow_w) || (out_albedo && !albedo) || (out_shading && !final_shading) || (out_normals && !normals) ||
        (out_depth && (!depth || !depth_range)))
        return GCFR_ERR_INVALID_ARGUMENT;
    ImagesArgs a{input_hwc, rendered, albedo, depth, depth_range, shadow_w, final_shading, normals, mask,
                 out_rendered, out_shadow, out_albedo, out_depth, out_shading, out_normals, mask_batch, L, H, W, mask_f32};
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL(inference_images_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)(B * L)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH;
}

extern "C" int gcfr_fix_border_u8(const uint8_t *img_hwc, const uint8_t *face_mask_u8, int32_t mask_batch, int32_t B, int32_t H,
                                  int32_t W, uint8_t *out_hwc, void *stream)
{
    if (!img_hwc || !face_mask_u8 || !out_hwc || img_hwc == out_hwc || B <= 0 || H <= 0 || W <= 0 || B > 65535 ||
        (mask_batch != 1 && mask_batch != B))
        return GCFR_ERR_INVALID_ARGUMENT;
    const size_t P = (size_t)H * W;
    hipLaunchKernelGGL(fix_border_kernel, dim3((unsigned)((P + 255) / 256), (unsigned)B), dim3(256), 0,
                       (hipStream_t)stream, img_hwc, face_mask_u8, mask_batch, H, W, out_hwc);
    return hipGetLastError() == hipSuccess ? GCFR_OK : GCFR_ERR_LAUNCH;
}
