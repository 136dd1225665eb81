// Ray-marched minimum point-to-line distance (the hot kernels), depth repack prepass, light preparation
// and the forward C entry points, gfx950.
//
// Replaces train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:357-363 and :371-515.  The reference
// materialises (N,2,H,W) f64 sample grids per image (2.8 GB of temporaries per 256x256 face); here
// one lane owns one pixel and walks its N samples in registers:
//   - one wavefront = one pixel tile (64 lanes); neighbouring lanes march neighbouring, nearly parallel
//     rays, so every gather instruction of the wave touches a footprint about the size of the tile ->
//     a handful of 128-B lines served by the CU's L1;
//   - the per-image working set (depth + mask, 320 KB at 256x256; 1.3 MB repacked) lives in L2, so
//     HBM traffic is compulsory only; the kernels are VALU / vector-memory-issue bound, not HBM bound;
//   - the sample table is wave-uniform: it is read with scalar loads (SMEM), costing no VALU;
//   - gathers use raw buffer loads (32-bit offsets off an SGPR descriptor: no 64-bit address VALU,
//     hardware range check instead of per-sample clamps).
// No MFMA: there is no dense contraction anywhere on this path.
//
// Two march kernels compute the same bits:
//   shadow_fwd_kernel       plain form (no workspace): four 4-byte depth gathers per ray-step;
//   shadow_fwd_quad_kernel  production form: repacked 2x2 texels, magic-number rint, exact skipping of
//                           masked work, optional fused shading epilogue (see its header comment).
#include "gcfr_march.hpp"

namespace gcfr {


// ----------------------------------------------------------------------------------------------
// light preparation, T8:357-363 / S1:332-336
// ----------------------------------------------------------------------------------------------
__device__ inline void light_prep_one(const float *__restrict__ light_raw, int i, int clamp_z,
                                      float clamp_min, float light_distance,
                                      float *__restrict__ unit_out, float *__restrict__ light_pt_out)
{
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

__global__ void light_prep_kernel(const float *__restrict__ light_raw, int n, int clamp_z,
                                  float clamp_min, float light_distance, float *__restrict__ unit_out,
                                  float *__restrict__ light_pt_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n)
        light_prep_one(light_raw, i, clamp_z, clamp_min, light_distance, unit_out, light_pt_out);
}


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
    rc.H = H;
    rc.W = W;
    rc.halfW = halfW;
    rc.halfH = halfH;
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
    rc.x64 = rc.x;
    rc.y64 = rc.y;
    rc.dx64 = finite_ray ? (double)rc.dx : 0.0;
    rc.dy64 = finite_ray ? (double)rc.dy : 0.0;

    // sqrt and the division by the per-pixel constant |BC| are monotone, so
    // min_k sqrt(S_k)/den == sqrt(min_k S_k)/den bit for bit: track the minimum of S over the
    // unmasked samples and finish once per pixel.
    float bestS = __builtin_inff();
    int besti = -1;
    // The running minimum it replaced last (distance-tie resolution: see first_tied_sample).
    float prevS = __builtin_inff();
    int prevk = -1;
    bool any_masked = false;

    for (int k = 0; k < N; ++k) {
        bool masked;
        const float S = ray_sample(rc, a.t_table[k], zr, mr, masked);  // t wave-uniform -> s_load
        any_masked |= masked;
        const bool take = !masked && (S < bestS);  // strict: first minimum wins (T8:514)
        prevS = take ? bestS : prevS;
        prevk = take ? besti : prevk;
        bestS = take ? S : bestS;
        besti = take ? k : besti;
    }

    const float den = __builtin_sqrtf(((rc.BCx * rc.BCx + rc.BCy * rc.BCy) + rc.BCz * rc.BCz) + kEps4);
    float d = __builtin_sqrtf(bestS) / den;  // +inf when every sample was masked
    if (a.argmin) {
        const bool tie = (prevk >= 0) && (__builtin_sqrtf(prevS) / den == d);
        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {  // wave-uniform branch: the helper shuffles
            const int first = first_tied_sample(rc, a.t_table, zr, mr, tie, prevk, den, d);
            besti = tie ? first : besti;
        }
    }
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

__device__ inline void build_zbounds_tile(int tile, int ls, int b, const float *__restrict__ depth,
                                          float4 *__restrict__ zb, int H, int W)
{
    const int ntw = (W >> ls) + 1, nth = (H >> ls) + 1;
    if (tile >= nth * ntw)
        return;
    const int lane = threadIdx.x & 63;
    const int ti = tile / ntw, tj = tile - ti * ntw;
    const int side = 2 << ls;
    const float *z = depth + (size_t)b * H * W;
    if (ls == 3 && ti >= 1 && tj >= 1 && (ti << 3) + 15 <= H && (tj << 3) + 15 <= W) {
        // Fast path (the common stride, a tile of proper cells only): each lane owns one 2x2 patch of its row's
        // quadrant, loads it once and uses it for both passes; no bounds tests, no reloads.
        const int q = lane >> 4, iy = (lane >> 2) & 3, ix = lane & 3;
        const int er = (ti << 3) + ((q >> 1) << 3) + (iy << 1), ec = (tj << 3) + ((q & 1) << 3) + (ix << 1);
        const float *p = z + (size_t)(er - 1) * W + (ec - 1);
        const float v00 = p[0], v01 = p[1], v10 = p[W], v11 = p[W + 1];
        const bool f00 = v00 - v00 == 0.0f, f01 = v01 - v01 == 0.0f, f10 = v10 - v10 == 0.0f, f11 = v11 - v11 == 0.0f;
        float cnt = ((f00 ? 1.0f : 0.0f) + (f01 ? 1.0f : 0.0f)) + ((f10 ? 1.0f : 0.0f) + (f11 ? 1.0f : 0.0f));
        float sz = ((f00 ? v00 : 0.0f) + (f01 ? v01 : 0.0f)) + ((f10 ? v10 : 0.0f) + (f11 ? v11 : 0.0f));
        cnt = row_sum_f32(cnt);
        sz = row_sum_f32(sz);
        const float n00 = lane_value(cnt, 15), n01 = lane_value(cnt, 31), n10 = lane_value(cnt, 47), n11 = lane_value(cnt, 63);
        float pa = 0.0f, pb = 0.0f;
        if (n00 > 0.0f && n01 > 0.0f && n10 > 0.0f && n11 > 0.0f) {
            const float m00 = lane_value(sz, 15) / n00, m01 = lane_value(sz, 31) / n01;
            const float m10 = lane_value(sz, 47) / n10, m11 = lane_value(sz, 63) / n11;
            const float ca = ((m01 + m11) - (m00 + m10)) * 0.0625f, cb = ((m10 + m11) - (m00 + m01)) * 0.0625f;
            if (ca - ca == 0.0f)
                pa = fminf(fmaxf(ca, -4.0f), 4.0f);
            if (cb - cb == 0.0f)
                pb = -fminf(fmaxf(cb, -4.0f), 4.0f);
        }
        const float X0 = (float)(ec - 1) - 0.5f * (float)W, Y0 = 0.5f * (float)H - (float)(er - 1);
        const float base = __builtin_fmaf(pa, X0, pb * Y0);  // the plane at the patch's first cell; +pa per column, -pb per row
        const float r00 = v00 - base, r01 = v01 - (base + pa), r10 = v10 - (base - pb), r11 = v11 - ((base + pa) - pb);
        const float lo = fminf(fminf(r00, r01), fminf(r10, r11));  // (fminf / fmaxf drop NaN cells)
        const float hi = fmaxf(fmaxf(r00, r01), fmaxf(r10, r11));
        const float wlo = f32_unsortable(wave_min_i32(f32_sortable(lo)));
        const float whi = -f32_unsortable(wave_min_i32(f32_sortable(-hi)));
        if (lane == 0)
            zb[(size_t)b * zb_slot(H, W) + tile] = make_float4(pa, pb, wlo, whi);
        return;
    }
    // pass 1: slopes from the means of the tile's four s x s quadrants (finite proper cells only).  Row q of
    // the wave (16 lanes) owns quadrant q = 2*qy + qx and strides over its cells, so the four sums come out of
    // DPP row reductions with no cross-row traffic.
    const int s = 1 << ls;
    const int q = lane >> 4, ql = lane & 15;
    const int qy = q >> 1, qx = q & 1;
    float cnt = 0.0f, sz = 0.0f;
    for (int e = ql; e < s * s; e += 16) {
        const int er = (ti << ls) + (qy << ls) + (e >> ls), ec = (tj << ls) + (qx << ls) + (e & (s - 1));
        if (er >= 1 && ec >= 1 && er <= H && ec <= W) {
            const float v = z[(size_t)(er - 1) * W + (ec - 1)];
            if (v - v == 0.0f) {
                cnt += 1.0f;
                sz += v;
            }
        }
    }
    cnt = row_sum_f32(cnt);
    sz = row_sum_f32(sz);
    const float n00 = lane_value(cnt, 15), n01 = lane_value(cnt, 31), n10 = lane_value(cnt, 47), n11 = lane_value(cnt, 63);
    float pa = 0.0f, pb = 0.0f;
    if (n00 > 0.0f && n01 > 0.0f && n10 > 0.0f && n11 > 0.0f) {  // all four quadrants populated (interior tiles)
        const float m00 = lane_value(sz, 15) / n00, m01 = lane_value(sz, 31) / n01;
        const float m10 = lane_value(sz, 47) / n10, m11 = lane_value(sz, 63) / n11;
        const float inv = 0.5f / (float)s;  // quadrant centres are s apart
        const float ca = ((m01 + m11) - (m00 + m10)) * inv, cb = ((m10 + m11) - (m00 + m01)) * inv;
        if (ca - ca == 0.0f)
            pa = fminf(fmaxf(ca, -4.0f), 4.0f);   // dz/dX: X grows with the column
        if (cb - cb == 0.0f)
            pb = -fminf(fmaxf(cb, -4.0f), 4.0f);  // dz/dY: Y falls with the row
    }
    // pass 2: residual extrema over every cell of the tile, wrap row / column included
    float lo = __builtin_inff(), hi = -__builtin_inff();
    for (int e = lane; e < side * side; e += 64) {
        const int er = (ti << ls) + (e >> (ls + 1)), ec = (tj << ls) + (e & (side - 1));
        if (er <= H && ec <= W) {
            const int r = er == 0 ? H - 1 : er - 1, c = ec == 0 ? W - 1 : ec - 1;
            const float X = (float)(ec - 1) - 0.5f * (float)W, Y = 0.5f * (float)H - (float)(er - 1);
            const float res = z[(size_t)r * W + c] - __builtin_fmaf(pa, X, pb * Y);
            lo = fminf(lo, res);  // NaN cells are ignored: a NaN sample never wins the minimum
            hi = fmaxf(hi, res);
        }
    }
    const float wlo = f32_unsortable(wave_min_i32(f32_sortable(lo)));
    const float whi = -f32_unsortable(wave_min_i32(f32_sortable(-hi)));
    if (lane == 0)
        zb[(size_t)b * zb_slot(H, W) + tile] = make_float4(pa, pb, wlo, whi);
}

// kZbTilesPerWave tiles per wave, one after the other: a block per four tiles was 273 blocks per 256 x 256 image, half of the
// prepass' workgroups -- and the prepass runs at the pace they are dispatched (see the repack job)
#ifndef GCFR_ZB_TILES_PER_WAVE
#define GCFR_ZB_TILES_PER_WAVE 2
#endif
constexpr int kZbTilesPerWave = GCFR_ZB_TILES_PER_WAVE;
__device__ inline void build_zbounds_block(int block, int b, const float *__restrict__ depth,
                                           float4 *__restrict__ zb, int H, int W, int N,
                                           const double *__restrict__ t_table, int group)
{
    const int ls = zb_log2_stride(H, W, N, t_table, group);
    if (block == 0 && threadIdx.x == 0)
        zb[(size_t)b * zb_slot(H, W) + zb_max_tiles(H, W) - 1] =
            make_float4(0.0f, 0.0f, -__builtin_inff(), __builtin_inff());
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    for (int t = 0; t < kZbTilesPerWave; ++t)
        build_zbounds_tile((block * kZbTilesPerWave + t) * 4 + wave, ls, b, depth, zb, H, W);
}

// diag[4]: {min (c + r), -max (c + r), min (c - r), -max (c - r)} over the non-zero cells: the mask's diagonal extents (the
// bounding OCTAGON together with the box; round 3)
__device__ inline void stat_mask_dword(uint32_t d, int r, int c, int &rmin, int &cmin, int &nrmax, int &ncmax, int &all_set,
                                       int (&diag)[4])
{
    const uint32_t nz = (d | ((d & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;  // bit 7 of every non-zero byte
    all_set &= (nz == 0x80808080u) ? 1 : 0;
    if (nz) {
        const int first = __builtin_ctz(nz) >> 3, last = (31 - __builtin_clz(nz)) >> 3;
        rmin = min(rmin, r);
        nrmax = min(nrmax, -r);
        cmin = min(cmin, c + first);
        ncmax = min(ncmax, -(c + last));
        diag[0] = min(diag[0], c + first + r);
        diag[1] = min(diag[1], -(c + last + r));
        diag[2] = min(diag[2], c + first - r);
        diag[3] = min(diag[3], -(c + last - r));
    }
}

__device__ inline void build_stats_block(int chunk, int b, const float *__restrict__ depth,
                                         const uint8_t *__restrict__ mask, int mask_batch, int H, int W,
                                         int *__restrict__ bbox, int *__restrict__ zrange, int *__restrict__ mones,
                                         int *__restrict__ diag_out, bool want_z, bool vec_ok)
{
    const int P = H * W;
    const int chunk_px = stat_chunk_px(H, W);  // kStatChunk or twice that
    const int p0 = chunk * chunk_px;
    const bool want_box = b < mask_batch;
    int rmin = kBBoxInit, cmin = kBBoxInit, nrmax = kBBoxInit, ncmax = kBBoxInit;
    int all_set = 1;  // every mask cell this lane saw is non-zero
    int diag[4] = {kBBoxInit, kBBoxInit, kBBoxInit, kBBoxInit};
    float zlo = __builtin_inff(), zhi = -__builtin_inff();  // fminf / fmaxf drop NaN cells (a NaN sample never wins)
    const float *z = depth + (size_t)b * P;
    const uint8_t *m = mask + (size_t)b * P;  // only dereferenced when want_box
    if (vec_ok) {  // W % 16 == 0 and 16-byte aligned planes: 16 pixels of one row per lane and step
        for (int part0 = 0; part0 < chunk_px; part0 += kStatChunk)
#pragma unroll
        for (int it = 0; it < kStatChunk / 4096; ++it) {
            const int i = p0 + part0 + it * 4096 + (int)threadIdx.x * 16;
            if (i < P) {
                if (want_z) {
                    const float4 *zp = (const float4 *)(z + i);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const float4 v = zp[q];
                        zlo = fminf(fminf(zlo, v.x), fminf(fminf(v.y, v.z), v.w));
                        zhi = fmaxf(fmaxf(zhi, v.x), fmaxf(fmaxf(v.y, v.z), v.w));
                    }
                }
                if (want_box) {
                    const uint4 mv = *(const uint4 *)(m + i);
                    const int r = i / W, c = i - r * W;
                    stat_mask_dword(mv.x, r, c, rmin, cmin, nrmax, ncmax, all_set, diag);
                    stat_mask_dword(mv.y, r, c + 4, rmin, cmin, nrmax, ncmax, all_set, diag);
                    stat_mask_dword(mv.z, r, c + 8, rmin, cmin, nrmax, ncmax, all_set, diag);
                    stat_mask_dword(mv.w, r, c + 12, rmin, cmin, nrmax, ncmax, all_set, diag);
                }
            }
        }
    } else {  // any width / alignment: one pixel per lane and step
        for (int e = threadIdx.x; e < chunk_px; e += 256) {
            const int i = p0 + e;
            if (i >= P)
                break;
            if (want_z) {
                const float v = z[i];
                zlo = fminf(zlo, v);
                zhi = fmaxf(zhi, v);
            }
            if (want_box) {
                if (m[i] != 0) {
                    const int r = i / W, c = i - r * W;
                    rmin = min(rmin, r);
                    cmin = min(cmin, c);
                    nrmax = min(nrmax, -r);
                    ncmax = min(ncmax, -c);
                    diag[0] = min(diag[0], c + r);
                    diag[1] = min(diag[1], -(c + r));
                    diag[2] = min(diag[2], c - r);
                    diag[3] = min(diag[3], -(c - r));
                } else {
                    all_set = 0;
                }
            }
        }
    }
    __shared__ int part[4][11];
    const int wv = threadIdx.x >> 6;
    const int v0 = wave_min_i32(rmin), v1 = wave_min_i32(cmin), v2 = wave_min_i32(nrmax), v3 = wave_min_i32(ncmax);
    const int v4 = wave_min_i32(f32_sortable(zlo)), v5 = wave_min_i32(f32_sortable(-zhi));
    const int v6 = __builtin_amdgcn_ballot_w64(all_set == 0) == 0ull ? 1 : 0;
    const int v7 = wave_min_i32(diag[0]), v8 = wave_min_i32(diag[1]), v9 = wave_min_i32(diag[2]), v10 = wave_min_i32(diag[3]);
    if ((threadIdx.x & 63) == 0) {
        part[wv][0] = v0;
        part[wv][1] = v1;
        part[wv][2] = v2;
        part[wv][3] = v3;
        part[wv][4] = v4;
        part[wv][5] = v5;
        part[wv][6] = v6;
        part[wv][7] = v7;
        part[wv][8] = v8;
        part[wv][9] = v9;
        part[wv][10] = v10;
    }
    __syncthreads();
    if (threadIdx.x < 11) {
        const int q = threadIdx.x;
        const int v = min(min(part[0][q], part[1][q]), min(part[2][q], part[3][q]));
        const size_t rec = (size_t)b * n_stat_chunks(H, W) + chunk;
        if (q < 4) {
            if (want_box)
                bbox[rec * 4 + q] = v;
        } else if (q < 6) {
            if (want_z)
                zrange[rec * 2 + (q - 4)] = v;
        } else if (q == 6) {
            if (want_box)
                mones[rec] = v;  // 1 iff every mask cell of the chunk is non-zero
        } else if (want_box) {
            diag_out[rec * 4 + (q - 7)] = v;
        }
    }
}

// Prepass, one launch.  Grid x = [depth-bounds tiles | statistics chunks | repack blocks], y = image:
//   (d) the depth-bounds tiles (head of the grid: their short dependent-load chains start first),
//   (c) per-chunk partial mask bounding boxes and depth ranges (build_stats_block),
//   (f) the mask bitmap of the LDS-staged march (build_bitmap_block), when that variant will run,
//   (a) the repack of depth into 2x2-neighbourhood texels; its first block also runs (b) the optional light
//       preparation and (e) the sample-table check.
// The mask as a bitmap (LDS-staged march): thread i of an image packs cells [32 i, 32 i + 32) into dword i.
__device__ inline void build_bitmap_block(int block, int b, const uint8_t *__restrict__ mask, int H, int W,
                                          uint32_t *__restrict__ bitmap)
{
    const int i = block * 256 + (int)threadIdx.x;
    if (i >= ((H * W) >> 5))
        return;
    const uint4 *m = (const uint4 *)(mask + (size_t)b * H * W) + 2 * (size_t)i;  // 32 bytes (W % 32 == 0: 16-byte aligned rows)
    const uint4 lo = m[0], hi = m[1];
    auto nib = [](uint32_t d) -> uint32_t {  // one bit per non-zero byte, byte k -> bit k
        const uint32_t nz = (d | ((d & 0x7f7f7f7fu) + 0x7f7f7f7fu)) & 0x80808080u;
        return ((nz >> 7) & 1u) | ((nz >> 14) & 2u) | ((nz >> 21) & 4u) | ((nz >> 28) & 8u);
    };
    const uint32_t bits = nib(lo.x) | (nib(lo.y) << 4) | (nib(lo.z) << 8) | (nib(lo.w) << 12) | (nib(hi.x) << 16) |
                          (nib(hi.y) << 20) | (nib(hi.z) << 24) | (nib(hi.w) << 28);
    bitmap[(size_t)b * (bitmap_stride_bytes(H, W) >> 2) + i] = bits;
}

// Horizon tables (round 3).  The march's early termination asks "is the ray, from here on, above everything it could still
// sample?" and answers with the image's depth maximum -- one number, so a ray that has cleared the nose keeps marching over
// the cheek until it is higher than the nose.  Sharper, and still one comparison: the maximum over the columns (rows) the
// REST of the ray can touch.  A ray runs monotonically in x and in y, so those are running maxima from its current column
// towards the side it is heading for: four tables per image,
//     col_pre[c] = max over columns <= c,  col_suf[c] = max over columns >= c,  row_pre / row_suf likewise
// (each kHorizonDim entries, centred, behind the image's depth-bounds records: see zb_slot), of
//     colmax[c] = max over the column's LIVE cells of max(depth, 0)  (rowmax likewise).
// A cell is live if an unmasked sample can read it: a sample's four bilinear corners lie within one cell of its rounded cell
// (T8:472-494), so the live cells are the mask's non-zero cells dilated by one -- here by up to one dword of columns more
// (the dilation works on ballots of "this lane's four mask bytes are not all zero"), a superset, which only makes the tables
// larger: still bounds.  The corners' wrap partners (column / row -1 reads the last one, the texel grid's last column / row
// pairs with the first) are in every entry: prefix tables include the last column / row, suffix tables the first.  0 is
// included because integral sample positions read z = 0 (see the depth-bound skip).  NaN cells are dropped (a NaN sample never
// wins), +inf stays +inf (never terminates).
// Parallel without a combining pass: the image's rows are split into kHorizonBands bands and a block per band computes the
// four tables OF ITS BAND ALONE (the column maxima over its rows; its own rows' maxima, zero elsewhere).  Maximum commutes with the running maxima, so the table proper is the element-wise maximum of the
// bands' tables -- which the march takes at look-up time: entry i is ONE float4 holding the four bands' values (one 16-byte
// gather).  Each block reads its quarter of the depth and mask planes once (float4 / dword per lane and row, all loads of a round in
// flight: GCFR_HORIZON_RB), reduces into LDS and scans there.  Needs W % 4 == 0, 16-byte aligned planes and H, W <= kHorizonDim.
__device__ inline void build_horizon_block(int k, int b, const float *__restrict__ depth, const uint8_t *__restrict__ mask,
                                           int mask_batch, int H, int W, float4 *__restrict__ zb)
{
    // column maxima over the band's rows / row maxima of the band's rows, as int bits (>= 0: integer order == float order),
    // and their running maxima from either end
    constexpr int kBandRows = kHorizonDim / kHorizonBands;
    __shared__ int s_col[kHorizonDim], s_cpre[kHorizonDim], s_csuf[kHorizonDim];
    __shared__ int s_row[kBandRows], s_rpre[kBandRows], s_rsuf[kBandRows];
    const int tid = (int)threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bandrows = (H + kHorizonBands - 1) / kHorizonBands, b_lo = min(H, k * bandrows), b_hi = min(H, b_lo + bandrows);
    const int nr = b_hi - b_lo;  // this band's rows (<= kBandRows)
    for (int i = tid; i < kHorizonDim; i += 256)
        s_col[i] = 0;
    if (tid < kBandRows)
        s_row[tid] = 0;
    __syncthreads();
    const size_t P = (size_t)H * W;
    const float *z = depth + (size_t)b * P;
    const uint8_t *m = mask + (size_t)(mask_batch == 1 ? 0 : b) * P;
    const int wrows = (nr + 3) >> 2, r_lo = min(b_hi, b_lo + wave * wrows), r_hi = min(b_hi, r_lo + wrows);  // this wave's rows
    const int segs = (W + 255) >> 8;                                                                        // 256 columns (64 lanes x 4) per segment
#ifndef GCFR_HORIZON_RB
#define GCFR_HORIZON_RB 8
#endif
    // rows per round: their RB + 2 mask dwords and RB depth float4 are all in flight before the first is used.  8: the most that
    // keeps the prepass kernel at 60 VGPRs = eight waves per SIMD -- with 16 (100 VGPRs, four waves) this job was a round
    // shorter and every OTHER job of the prepass slower: -6 % on the bench with four batches in flight
    constexpr int RB = GCFR_HORIZON_RB;
    for (int sg = 0; sg < segs; ++sg) {
        const int c = (sg << 8) + (lane << 2);
        const bool in_w = c < W;
        const int cc = in_w ? c : 0;  // (loads stay inside the plane; their values are dropped)
        // dilation across the segment's edges is not looked up: with more than one segment the edge lanes are always live
        // -- and so is the image's last dword of columns, the wrap partner of column 0 (below)
        const int last_lane = min(63, ((W - (sg << 8)) >> 2) - 1);  // this segment's last lane inside the image
        const unsigned long long edge = segs > 1 ? (0x8000000000000001ull | (1ull << last_lane)) : 0ull;
        // The dilation WRAPS where the reference's gathers do (index -1 == last, T8:488-491): a sample whose rounded cell lies in
        // column 0 (row 0) reads column W-1 (row H-1) as its left (upper) bilinear corner, with weight 1e-4 when it sits on the
        // image's edge.  So a non-zero mask cell in column 0 makes the last column live in its three rows, and one in row 0 makes
        // row H-1 live in its columns (round 4, advisor r03: a mask touching only the left / top edge left the wrap partner out
        // of the tables, and a large masked-out depth there was not covered by the trailing loop's cap).
        const unsigned long long wrap_bit = GCFR_M(14, true ||, ) segs > 1 ? 0ull : (1ull << last_lane);
        float4 cm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);  // running maxima of this lane's four columns
        for (int r0 = r_lo; r0 < r_hi; r0 += RB) {
            uint32_t md[RB + 2];
            float4 dv[RB];
#pragma unroll
            for (int j = 0; j < RB + 2; ++j) {
                const int r = r0 - 1 + j;
                md[j] = *(const uint32_t *)(m + (size_t)(r == H ? 0 : min(max(r, 0), H - 1)) * W + cc);  // (row H: row 0, the wrap)
            }
#pragma unroll
            for (int j = 0; j < RB; ++j)
                dv[j] = *(const float4 *)(z + (size_t)min(r0 + j, H - 1) * W + cc);
            unsigned long long bits[RB + 2];  // lanes whose four mask cells of the row are not all zero
#pragma unroll
            for (int j = 0; j < RB + 2; ++j) {
                const int r = r0 - 1 + j;
                bits[j] = __builtin_amdgcn_ballot_w64(in_w && r >= 0 && r GCFR_M(14, <, <=) H && md[j] != 0);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int r = r0 + j;
                const unsigned long long v3 = bits[j] | bits[j + 1] | bits[j + 2];
                const unsigned long long live = GCFR_M(20, bits[j + 1] | edge, v3 | (v3 << 1) | (v3 >> 1) | edge | ((v3 & 1ull) ? wrap_bit : 0ull));
                float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                if (in_w && r < r_hi && ((live >> lane) & 1ull))  // (fmaxf drops NaN)
                    v = make_float4(fmaxf(dv[j].x, 0.0f), fmaxf(dv[j].y, 0.0f), fmaxf(dv[j].z, 0.0f), fmaxf(dv[j].w, 0.0f));
                cm = make_float4(fmaxf(cm.x, v.x), fmaxf(cm.y, v.y), fmaxf(cm.z, v.z), fmaxf(cm.w, v.w));
                if (r < r_hi) {  // (wave-uniform)
                    const float rm = fmaxf(fmaxf(v.x, v.y), fmaxf(v.z, v.w));
                    const int wmax = -wave_min_i32(-__builtin_bit_cast(int, rm));  // (bits of a float >= 0)
                    if (lane == 0 && wmax != 0)
                        atomicMax(&s_row[r - b_lo], wmax);  // (one writer per row and segment; max across the segments)
                }
            }
        }
        if (in_w) {
            atomicMax(&s_col[c + 0], __builtin_bit_cast(int, cm.x));
            atomicMax(&s_col[c + 1], __builtin_bit_cast(int, cm.y));
            atomicMax(&s_col[c + 2], __builtin_bit_cast(int, cm.z));
            atomicMax(&s_col[c + 3], __builtin_bit_cast(int, cm.w));
        }
    }
    __syncthreads();
    // running maxima from either end, one wave per scan: columns prefix / suffix, rows prefix / suffix; a lane owns E consecutive entries
    {
        const bool rows = wave >= 2, rev = (wave & 1) != 0;
        const int n = rows ? nr : W;
        const int *src = rows ? s_row : s_col;
        int *dst = rows ? (rev ? s_rsuf : s_rpre) : (rev ? s_csuf : s_cpre);
        const int E = (n + 63) >> 6, i0 = lane * E;
        int run = 0;
        for (int e = 0; e < E; ++e) {
            const int i = i0 + e;  // position along the scan's direction
            if (i < n) {
                run = max(run, src[rev ? n - 1 - i : i]);
                dst[rev ? n - 1 - i : i] = run;
            }
        }
        int incl = run;  // inclusive scan of the lanes' totals, then the exclusive one
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(incl, off);
            if (lane >= off)
                incl = max(incl, o);
        }
        int excl = __shfl_up(incl, 1);
        if (lane == 0)
            excl = 0;
        for (int e = 0; e < E; ++e) {
            const int i = i0 + e;
            if (i < n) {
                int *p = &dst[rev ? n - 1 - i : i];
                *p = max(*p, excl);
            }
        }
    }
    __syncthreads();
    // centred tables: entry kHorizonDim/2 + (j - N/2) belongs to column / row j of the N the image has; entries outside repeat
    // the nearest one.  Component k of the entry's float4 is this band's value.  The wrap partners: every prefix entry includes
    // the image's last column / row, every suffix entry its first (of THIS band's values).
    // (Round 5) The SLACK of the march's look-up lives here, in the tables: the march asks with the cell its NEXT sample's f32
    // position falls into; that sample's bilinear corners reach one cell back (u = s - 0.0001 floors to the cell before an integral
    // s, T8:480-487) and one cell on, and the f32 position may be one cell off the f64 one -- so the suffix entry of cell c covers
    // the columns >= c - kHzBack, the prefix entry the columns <= c + kHzAhead.  (Rounds 3-4 added the slack in the march's index
    // arithmetic; as part of the table it is part of what tests/test_gpu_horizon.py pins, entry by entry.)
    constexpr int S = kHorizonDim;
    constexpr int kHzBack = GCFR_M(21, 0, 2), kHzAhead = GCFR_M(21, 0, 3);
    float *out = (float *)(zb + (size_t)b * zb_slot(H, W) + zb_stride(H, W)) + k;
    {   // columns: this band's column maxima over all W columns
        const int add_pre = GCFR_M(13, 0, s_col[W - 1]), add_suf = GCFR_M(13, 0, s_col[0]);
        for (int i = tid; i < S; i += 256) {
            const int c0 = i - S / 2 + W / 2;
            const int jp = min(max(c0 + kHzAhead, 0), W - 1), js = min(max(c0 - kHzBack, 0), W - 1);
            out[(size_t)i * 4] = __builtin_bit_cast(float, max(s_cpre[jp], add_pre));
            out[(size_t)(S + i) * 4] = __builtin_bit_cast(float, max(s_csuf[js], add_suf));
        }
    }
    {   // rows: this band's own rows' maxima, zero elsewhere
        const int total = nr > 0 ? s_rpre[nr - 1] : 0;
        const int add_pre = (GCFR_M(13, false &&, ) nr > 0 && b_hi == H) ? s_row[nr - 1] : 0;  // the band that holds the image's last row
        const int add_suf = (GCFR_M(13, false &&, ) nr > 0 && b_lo == 0) ? s_row[0] : 0;       // ... its first row
        for (int i = tid; i < S; i += 256) {
            const int r0 = i - S / 2 + H / 2;
            const int jp = min(max(r0 + kHzAhead, 0), H - 1) - b_lo, js = min(max(r0 - kHzBack, 0), H - 1) - b_lo;  // positions among this band's rows (may lie outside)
            const int pre = nr <= 0 ? 0 : (jp < 0 ? 0 : (jp >= nr ? total : s_rpre[jp]));
            const int suf = nr <= 0 ? 0 : (js < 0 ? total : (js >= nr ? 0 : s_rsuf[js]));
            out[(size_t)(2 * S + i) * 4] = __builtin_bit_cast(float, max(pre, add_pre));
            out[(size_t)(3 * S + i) * 4] = __builtin_bit_cast(float, max(suf, add_suf));
        }
    }
}

__global__ __launch_bounds__(256) void build_quad_kernel(const float *__restrict__ depth,
                                                         float4 *__restrict__ quad, int H, int W,
                                                         PrepassLights pl,
                                                         const uint8_t *__restrict__ mask, int mask_batch,
                                                         int *__restrict__ bbox, int *__restrict__ zrange,
                                                         int *__restrict__ mones,
                                                         float4 *__restrict__ zb, int zb_blocks, int stat_blocks,
                                                         int want_z, int vec_ok, int N,
                                                         const double *__restrict__ t_table, int group,
                                                         int *__restrict__ tflag, uint32_t *__restrict__ bitmap,
                                                         int bitmap_blocks, int hz_blocks, int *__restrict__ diag)
{
    const int Wp = W + 1, Hp = H + 1;
    const int b = blockIdx.y;
    // (-DGCFR_PREPASS_JOBS=<bit mask>: which jobs run -- horizon 1, bounds 2, statistics 4, bitmap 8, repack 16; timing experiments
    //  only, tools/prepass_jobs.py: the workspace is garbage without all of them)
#ifdef GCFR_PREPASS_JOBS
#define GCFR_JOB(bit) (((GCFR_PREPASS_JOBS) >> (bit)) & 1)
#else
#define GCFR_JOB(bit) 1
#endif
    if ((int)blockIdx.x < hz_blocks) {  // (head of the grid: the longest job of the prepass; one block per image and row band)
        if (GCFR_JOB(0))
            build_horizon_block((int)blockIdx.x, b, depth, mask, mask_batch, H, W, zb);
        return;
    }
    const int bx = (int)blockIdx.x - hz_blocks;
    if (bx < zb_blocks) {
        if (GCFR_JOB(1))
            build_zbounds_block(bx, b, depth, zb, H, W, N, t_table, group);
        return;
    }
    if (bx < zb_blocks + stat_blocks) {
        if (GCFR_JOB(2))
            build_stats_block(bx - zb_blocks, b, depth, mask, mask_batch, H, W, bbox, zrange, mones, diag,
                              want_z != 0, vec_ok != 0);
        return;
    }
    if (bx < zb_blocks + stat_blocks + bitmap_blocks) {
        if (GCFR_JOB(3) && b < mask_batch)
            build_bitmap_block(bx - zb_blocks - stat_blocks, b, mask, H, W, bitmap);
        return;
    }
    const int qb = bx - zb_blocks - stat_blocks - bitmap_blocks;
    if (!GCFR_JOB(4))
        return;
#ifndef GCFR_QUAD_PER_THREAD
#define GCFR_QUAD_PER_THREAD 4
#endif
    constexpr int QP = GCFR_QUAD_PER_THREAD;  // texels per thread of the repack job
    if (qb == 0 && b == 0 && threadIdx.x < 64) {
        // Is the sample table what the march's pruning / skipping reasons about -- increasing, inside [0, 1]
        // (every sample between the pixel and its end point) and uniform to 0.1 %?  One wave checks, once per launch.
        bool ok = (N >= 2) && (t_table[0] >= 0.0) && (t_table[N - 1] <= 1.0);
        if (ok) {
            const double step = (t_table[N - 1] - t_table[0]) / (double)(N - 1);
            ok = step > 0.0;
            for (int k = threadIdx.x; k < N - 1; k += 64) {
                const double dk = t_table[k + 1] - t_table[k];
                ok = ok && (dk > 0.0) && (fabs(dk - step) <= 1e-3 * step);
            }
        }
        const bool all_ok = __builtin_amdgcn_ballot_w64(!ok) == 0ull;
        if (threadIdx.x == 0) {
            tflag[0] = all_ok ? 1 : 0;
            // What every march tile derives from the sample table alone, once per launch instead of once per tile (round 3:
            // gfx950 has no scalar float unit, so this wave-uniform arithmetic -- an f64 division among it -- ran on the
            // VECTOR unit in every tile's prologue, ~3 % of the march's VALU instructions): the bounds grid's stride and
            // whether a group's footprint always fits one tile, max |t|, and t -> sample-index conversion of the pruning.
            bool fits = false;
            const int ls = zb_log2_stride(H, W, N, t_table, group, &fits);
            tflag[kTfStride] = ls | (fits ? 0x100 : 0);
            const float t_first = (N >= 1) ? (float)t_table[0] : 0.0f, t_last = (N >= 1) ? (float)t_table[N - 1] : 0.0f;
            tflag[kTfTabs] = __builtin_bit_cast(int, fmaxf(fabsf(t_first), fabsf(t_last)));
            tflag[kTfTfirst] = __builtin_bit_cast(int, t_first);
            tflag[kTfInvDt] = __builtin_bit_cast(int, (float)(N - 1) * __builtin_amdgcn_rcpf(t_last - t_first));
        }
    }
    if (pl.light_raw && qb == 0) {
        for (int l = threadIdx.x; l < pl.L; l += blockDim.x)
            light_prep_one(pl.light_raw, b * pl.L + l, pl.clamp_z, pl.clamp_min, pl.light_distance,
                           pl.unit_out, pl.light_pt_out);
    }
    // QP texels per thread, 256 apart (coalesced), all their loads in flight before the first store: a block per 256 texels was
    // 258 blocks per 256 x 256 image -- at B = 128 a third of a million workgroups, and the prepass ran at the dispatcher's pace
    const float *z = depth + (size_t)b * H * W;
    float4 q[QP];
#pragma unroll
    for (int j = 0; j < QP; ++j) {
        const int i = min((qb * QP + j) * 256 + (int)threadIdx.x, Hp * Wp - 1);
        const int rp = i / Wp, cp = i - rp * Wp;
        const int r = rp - 1, c = cp - 1;
        const int r0 = r < 0 ? H - 1 : r, c0 = c < 0 ? W - 1 : c;
        const int r1 = (r + 1 >= H) ? 0 : r + 1, c1 = (c + 1 >= W) ? 0 : c + 1;
        q[j] = make_float4(z[(size_t)r0 * W + c0], z[(size_t)r0 * W + c1], z[(size_t)r1 * W + c0], z[(size_t)r1 * W + c1]);
    }
#pragma unroll
    for (int j = 0; j < QP; ++j) {
        const int i = (qb * QP + j) * 256 + (int)threadIdx.x;
        if (i < Hp * Wp)
            quad[(size_t)b * Hp * Wp + i] = q[j];
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

extern "C" const char *gcfr_version(void)
{
#if defined(GCFR_AUDIT)
    return "gcfr-hip 0.5.0 gfx950 +counters +audit";
#elif defined(GCFR_COUNTERS)
    return "gcfr-hip 0.5.0 gfx950 +counters";
#else
    return "gcfr-hip 0.5.0 gfx950";
#endif
}

extern "C" int32_t gcfr_abi_version(void)
{
    return GCFR_ABI_VERSION;
}

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

// Per-call knobs, resolved from gcfr_options (NULL = defaults).  Measured on MI355X, B=8 x 256^2 x 160
// (profiles/): tile 2x32 > 4x16 > 8x8 > 1x64 without the depth-bound skip,
// 8x8 >= 4x16 > 2x32 > 1x64 with it; an XCD-affine block map is 12 % slower; f64 texels (32-B gathers) 25 % slower.
struct Knobs {
    int tile_w = 0;      // pixels per tile row: 8, 16, 32 or 64 (tile = 64/tile_w rows); 0 = auto
    int group = 4;       // samples per group (skip granularity / gathers in flight): 1, 2 or 4
    int ksplit = -1;     // sample-range split over the 4 waves of a workgroup: 0 off, 1 on, -1 auto
    int zbound = 1;      // depth-bound group skip (exact): 1 on, 0 off
    int lds_stage = -1;  // mask bitmap + bounds records of the workgroup's image in LDS: 0 off, 1 on (where the shape allows), -1 auto
    int pixels = 0;      // 1: pixels outside the mask are not marched (gcfr_options.pixels; the one knob that changes results)
    int phase = 0;       // 0: prepass + march, 1: the prepass only, 2: the march only (gcfr_options.phase)
    hipEvent_t ev_start = nullptr, ev_stop = nullptr;
    unsigned long long *counters = nullptr;
};

// gcfr_options.phase == 1 (the prepass alone), read only from a struct whose size has been checked: a caller built against
// another revision of the struct is refused by resolve_options() below, never interpreted field by field
static inline bool prepass_only(const gcfr_options *opt)
{
    return opt && opt->struct_size == sizeof(gcfr_options) && opt->phase == 1;
}

static_assert(sizeof(gcfr_options) == 56, "gcfr_options: ABI revision 6 layout (include/gcfr.h); bump GCFR_ABI_VERSION with it");

static int resolve_options(const gcfr_options *opt, Knobs &k)
{
    if (!opt)
        return GCFR_OK;
    if (opt->struct_size != sizeof(gcfr_options))
        return GCFR_ERR_INVALID_ARGUMENT;
    const int tw = opt->tile_w, g = opt->group;
    if ((tw != 0 && tw != 8 && tw != 16 && tw != 32 && tw != 64) || (g != 0 && g != 1 && g != 2 && g != 4) ||
        opt->ksplit < -1 || opt->ksplit > 1 || opt->depth_bound_skip < -1 || opt->depth_bound_skip > 1 ||
        opt->lds_stage < -1 || opt->lds_stage > 1 || opt->pixels < -1 || opt->pixels > 1 || opt->phase < -1 || opt->phase > 2)
        return GCFR_ERR_INVALID_ARGUMENT;
    k.tile_w = tw;
    k.group = g ? g : 4;
    k.ksplit = opt->ksplit;
    k.zbound = opt->depth_bound_skip < 0 ? 1 : opt->depth_bound_skip;
    k.lds_stage = opt->lds_stage;
    k.pixels = opt->pixels == 1 ? 1 : 0;
    k.phase = opt->phase < 0 ? 0 : opt->phase;
    k.ev_start = (hipEvent_t)opt->event_start;
    k.ev_stop = (hipEvent_t)opt->event_stop;
    k.counters = (unsigned long long *)opt->counters;
    return GCFR_OK;
}

extern "C" void gcfr_options_default(gcfr_options *opt)
{
    if (!opt)
        return;
    *opt = gcfr_options{};
    opt->struct_size = (uint32_t)sizeof(gcfr_options);
    opt->ksplit = opt->depth_bound_skip = opt->lds_stage = -1;
    opt->pixels = opt->phase = 0;
}

// workspace layout: [quad texels | partial boxes | per image: depth-bounds records, horizon tables | partial depth ranges | tflag | all-ones flags | mask bitmaps | partial diagonal extents]
extern "C" size_t gcfr_shadow_workspace_bytes(int32_t B, int32_t H, int32_t W)
{
    if (B <= 0 || H <= 0 || W <= 0)
        return 0;
    const size_t n_stat = (size_t)n_stat_chunks(H, W);
    return (size_t)B * (size_t)(H + 1) * (size_t)(W + 1) * sizeof(float4) + (size_t)B * n_stat * 4 * sizeof(int) +
           (size_t)B * (size_t)zb_slot(H, W) * sizeof(float4) + (size_t)B * n_stat * 2 * sizeof(int) +
           (kQueueSlot + 1) * sizeof(int) + 12 + (size_t)B * n_stat * sizeof(int) + 16 +
           (((W & 31) == 0) ? (size_t)B * (size_t)bitmap_stride_bytes(H, W) : 0) +  // mask bitmaps (LDS-staged march)
           16 + (size_t)B * n_stat * 4 * sizeof(int);                                  // partial diagonal extents of the masks
}

// The march's translation unit for a tile shape (gcfr_march_unit.hip, one compilation per shape)
static MarchUnitFn march_unit(int tile_w, int group)
{
#ifndef GCFR_FAST_BUILD   // (development builds have the default shape only: tools/build_variant.sh ... -DGCFR_FAST_BUILD)
    switch (tile_w * 8 + group) {
    case 16 * 8 + 2: return launch_march_16_2;
    case 16 * 8 + 1: return launch_march_16_1;
    case 8 * 8 + 4: return launch_march_8_4;
    case 8 * 8 + 2: return launch_march_8_2;
    case 8 * 8 + 1: return launch_march_8_1;
    case 32 * 8 + 4: return launch_march_32_4;
    case 32 * 8 + 2: return launch_march_32_2;
    case 32 * 8 + 1: return launch_march_32_1;
    case 64 * 8 + 4: return launch_march_64_4;
    case 64 * 8 + 2: return launch_march_64_2;
    case 64 * 8 + 1: return launch_march_64_1;
    default: break;
    }
#endif
    (void)tile_w;
    (void)group;
    return launch_march_16_4;
}

static void launch_quad(ShadowQuadArgs a, bool even_half, bool want_argmin, int total_bl, Schedule sch,
                        const Knobs &kn, hipStream_t st, unsigned lds_bytes, int tile_w)
{
    const MarchUnitFn unit = march_unit(tile_w, kn.group);
    if (kn.ev_start)
        (void)hipEventRecord(kn.ev_start, st);
    const unsigned gx = (sch == kKSplit) ? (unsigned)a.tiles_x : (unsigned)((a.tiles_x + 3) / 4);  // (kGrid, kGridLds: four tiles per workgroup)
    for (int z0 = 0; z0 < total_bl; z0 += 65535) {  // grid z is limited to 65535 (image, light) pairs per launch
        a.bl_offset = z0;
        const unsigned gz = (unsigned)((total_bl - z0) < 65535 ? (total_bl - z0) : 65535);
        unit(a, even_half, want_argmin, sch, dim3(gx, (unsigned)a.tiles_y, gz), st, lds_bytes);
    }
    if (kn.ev_stop)
        (void)hipEventRecord(kn.ev_stop, st);
}

struct FusedShade {  // operands of the fused shading epilogue; rendered == nullptr: march only
    NormalsArgs nrm = {};            // used when normals == nullptr
    float *normals_out = nullptr;
    const float *normals = nullptr, *albedo = nullptr, *ambient = nullptr;
    float *shadow_w = nullptr, *full = nullptr, *final_shading = nullptr, *rendered = nullptr;
    float intensity = 0.0f;
    PrepassLights lights;  // light_raw != nullptr: the prepass also writes unit / light_pt
};

static int shadow_fwd_impl(const float *depth, const uint8_t *mask_u8, int32_t mask_batch,
                           const float *light_pt, int32_t B, int32_t L, int32_t H, int32_t W, int32_t N,
                           const double *t_table, float bonus, const float *bonus_box, float *min_dist,
                           int32_t *argmin, void *workspace, size_t workspace_bytes, void *stream,
                           const FusedShade &fs, const gcfr_options *opt)
{
    const bool pre_only = prepass_only(opt);  // (the prepass alone: the march's operands may still be missing)
    if (!depth || !mask_u8 || !light_pt || !t_table || (!min_dist && !pre_only))
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0 || N <= 0 || N > 4096 || H < 2 || W < 2 || H > 4096 || W > 4096 ||
        (H & 1) || (W & 1) || (mask_batch != 1 && mask_batch != B))
        return GCFR_ERR_INVALID_ARGUMENT;
    if (bonus != 0.0f && !bonus_box)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (workspace && (workspace_bytes < gcfr_shadow_workspace_bytes(B, H, W) || ((uintptr_t)workspace & 15u)))
        return GCFR_ERR_INVALID_ARGUMENT;  // (float4 records: 16-byte alignment)
    Knobs kn;
    if (resolve_options(opt, kn) != GCFR_OK)
        return GCFR_ERR_INVALID_ARGUMENT;
    if (kn.pixels == 1 && (!workspace || (!argmin && !pre_only)))
        return GCFR_ERR_INVALID_ARGUMENT;  // pixels = mask lives in the workspace path's training (argmin) march
    if (kn.phase != 0 && !workspace)
        return GCFR_ERR_INVALID_ARGUMENT;  // the two halves only exist where there is a prepass

    // auto tile shape (measured): with the depth-bound skip compact tiles win (the lanes of a wave
    // agree more often).  16x4 everywhere: on the smooth bench faces it ties with 8x8 at 256 px (1690 vs 1688 G
    // ray-steps/s) and wins above, and it degrades more gracefully when the bounds stop helping -- rough depth
    // (+15 % at noise 400, level with the kernel without bounds), all-ones masks (+4 %).  Without bounds 32x2 streams best.
    const int tile_auto = (kn.zbound && N >= 2) ? 16 : 32;
#ifdef GCFR_FAST_BUILD   // (development builds instantiate 16 x 4 tiles and groups of four only)
    kn.tile_w = 16;
    kn.group = 4;
#endif
    const int TILE_W = workspace ? (kn.tile_w ? kn.tile_w : tile_auto) : 16, TILE_H = 64 / TILE_W, WAVES = 4;
    const int tiles_x = (W + TILE_W - 1) / TILE_W, tiles_y = (H + TILE_H - 1) / TILE_H;
    const int quads_x = (tiles_x + WAVES - 1) / WAVES;
    const int quads_per_image = quads_x * tiles_y;
    const long long blocks = (long long)B * L * quads_per_image;
    const long long tiles_total = (long long)B * L * tiles_x * tiles_y;
    if (blocks > 0x7fffffffLL || tiles_total > 0x7fffffffLL || (long long)B * L > 0x7fffffffLL / 4)
        return GCFR_ERR_INVALID_ARGUMENT;
    float bx[4] = {0.0f, -1.0f, 0.0f, -1.0f};
    if (bonus_box)
        for (int i = 0; i < 4; ++i)
            bx[i] = bonus_box[i];
    hipStream_t st = (hipStream_t)stream;

    if (workspace) {
        // prepass (2x2 neighbourhood grid, statistics, depth bounds: see build_quad_kernel), then the march
        const int texels = (H + 1) * (W + 1);
        const size_t n_stat = (size_t)n_stat_chunks(H, W);
        int *bbox = (int *)((char *)workspace + (size_t)B * texels * sizeof(float4));
        float4 *zb = (float4 *)((char *)bbox + (size_t)B * n_stat * 4 * sizeof(int));
        int *zrange = (int *)(zb + (size_t)B * zb_slot(H, W));  // (B, n_stat, 2)
        int *tflag = zrange + (size_t)B * n_stat * 2;                 // [0] table flag
        int *mones = tflag + kQueueSlot + 4;                          // (B, n_stat) all-ones flags of the mask chunks
        uint32_t *bitmap = (uint32_t *)(((uintptr_t)(mones + (size_t)B * n_stat) + 15u) & ~(uintptr_t)15u);  // (MB, stride) 16-B aligned
        const bool use_zb = kn.zbound && N >= 2;
        int *diag = (int *)(((uintptr_t)((char *)bitmap + (((W & 31) == 0) ? (size_t)B * (size_t)bitmap_stride_bytes(H, W) : 0)) + 15u) &
                            ~(uintptr_t)15u);  // (MB, n_stat, 4)
        // Schedule.  Tiny launches (<= 2048 tiles, B <= 2 at 256^2): split every tile's sample range over the 4
        // waves of its workgroup (finer, more uniform pieces; a quarter-range wave starts the depth-bound skip
        // without a running minimum, so it loses from B = 4 up).  Otherwise the grid: one wave per tile -- with the
        // image's mask bitmap and bounds records staged in LDS where they fit beside five other workgroups of the CU
        // (26 KiB at 256 x 256; not at 512 x 512) and the rows are whole bitmap dwords.
        const bool own = kn.pixels == 1;  // pixels = mask: the grid schedule's own kernel
        const bool ksplit = !own && ((kn.ksplit < 0) ? (tiles_total <= 2048 && N >= 16) : (kn.ksplit == 1));
        const unsigned lds_bytes = (unsigned)bitmap_stride_bytes(H, W) + (use_zb ? (unsigned)zb_stride(H, W) * 16u : 0u);
        const bool lds_fits = ((W & 31) == 0) && (((uintptr_t)mask_u8 & 15u) == 0) && lds_bytes <= 26u * 1024u &&
                              TILE_W == 16 && kn.group == 4;
#ifndef GCFR_LDS_STAGE_AUTO
#define GCFR_LDS_STAGE_AUTO 0
#endif
        const bool lds_stage = !own && !ksplit && lds_fits && (kn.lds_stage < 0 ? (GCFR_LDS_STAGE_AUTO != 0) : (kn.lds_stage == 1));
        const Schedule sch = own ? kGridOwn : (ksplit ? kKSplit : (lds_stage ? kGridLds : kGrid));
        const int quad_blocks = (texels + 256 * GCFR_QUAD_PER_THREAD - 1) / (256 * GCFR_QUAD_PER_THREAD);
        const int zb_blocks = use_zb ? (zb_max_tiles(H, W) + 4 * kZbTilesPerWave - 1) / (4 * kZbTilesPerWave) : 0;  // sized for the finest stride
        const int bitmap_blocks = lds_stage ? ((H * W) / 32 + 255) / 256 : 0;
        // horizon tables: for the trailing loop of the grid schedule's bounds-skipping march (not the k-split's quarter
        // ranges, not the LDS-staged variant), where the shape and the planes' alignment allow vector loads
        const bool horizon = (GCFR_HORIZON != 0) && use_zb && (sch == kGrid || sch == kGridOwn) && hz_shape_ok(H, W) && (((uintptr_t)depth & 15u) == 0) &&
                             (((uintptr_t)mask_u8 & 3u) == 0);
        const int hz_blocks = horizon ? kHorizonBands : 0;
        const int vec_ok = ((W & 15) == 0) && (((uintptr_t)depth & 15u) == 0) && (((uintptr_t)mask_u8 & 15u) == 0);
        if (kn.phase != 2)  // (phase 2: a phase-1 call with the same arguments has filled the workspace)
            hipLaunchKernelGGL(build_quad_kernel, dim3(hz_blocks + zb_blocks + (int)n_stat + bitmap_blocks + quad_blocks, B), dim3(256), 0, st,
                               depth, (float4 *)workspace, H, W, fs.lights, mask_u8, mask_batch, bbox, zrange, mones, zb, zb_blocks,
                               (int)n_stat, use_zb ? 1 : 0, vec_ok, N, t_table, kn.group, tflag, bitmap, bitmap_blocks, hz_blocks, diag);
        if (kn.phase == 1)
            return launch_status();
        ShadowQuadArgs a = {};
        a.zb = use_zb ? zb : nullptr;
        a.zrange = zrange;
        a.mones = mones;
        a.tflag = tflag;
        a.depth = depth;
        a.quad = (const float4 *)workspace;
        a.bbox = bbox;
        a.diag = diag;
        a.mask = mask_u8;
        a.bitmap = bitmap;
        a.hz_off = horizon ? zb_stride(H, W) * 16 : -1;
        a.light_pt = light_pt;
        a.t_table = t_table;
        a.counters = kn.counters;
        a.mask_batch = mask_batch;
        a.B = B;
        a.L = L;
        a.H = H;
        a.W = W;
        a.N = N;
        a.tiles_x = tiles_x;
        a.tiles_y = tiles_y;
        a.bl_offset = 0;
        a.epi.min_dist = min_dist;
        a.epi.argmin = argmin;
        a.epi.bonus = bonus;
        a.epi.bx_lo = bx[0];
        a.epi.bx_hi = bx[1];
        a.epi.by_lo = bx[2];
        a.epi.by_hi = bx[3];
        a.epi.normals = fs.normals;
        a.epi.albedo = fs.albedo;
        a.epi.ambient = fs.ambient;
        a.epi.shadow_w = fs.shadow_w;
        a.epi.full = fs.full;
        a.epi.final_shading = fs.final_shading;
        a.epi.rendered = fs.rendered;
        a.epi.intensity = fs.intensity;
        a.epi.nrm = fs.nrm;
        a.epi.nrm.depth = depth;
        a.epi.nrm.H = H;
        a.epi.nrm.W = W;
        a.epi.normals_out = fs.normals_out;
        const bool even_half = (((W / 2) & 1) == 0) && (((H / 2) & 1) == 0);
        const bool want = argmin != nullptr;
        launch_quad(a, even_half, want, B * L, sch, kn, st, lds_bytes, TILE_W);
        return launch_status();
    }

    if (fs.rendered)
        return GCFR_ERR_INVALID_ARGUMENT;  // the fused epilogue lives in the workspace kernel only
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
    a.tiles_x = tiles_x;
    a.tiles_per_image = quads_per_image;
    a.bonus = bonus;
    a.bx_lo = bx[0];
    a.bx_hi = bx[1];
    a.by_lo = bx[2];
    a.by_hi = bx[3];
    if (kn.ev_start)
        (void)hipEventRecord(kn.ev_start, st);
    hipLaunchKernelGGL(shadow_fwd_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, st, a);
    if (kn.ev_stop)
        (void)hipEventRecord(kn.ev_stop, st);
    return launch_status();
}

extern "C" int gcfr_shadow_fwd(const float *depth, const uint8_t *mask_u8, int32_t mask_batch,
                               const float *light_pt, int32_t B, int32_t L, int32_t H, int32_t W,
                               int32_t N, const double *t_table, float bonus,
                               const float *bonus_box, float *min_dist, int32_t *argmin,
                               void *workspace, size_t workspace_bytes, void *stream,
                               const gcfr_options *opt)
{
    return shadow_fwd_impl(depth, mask_u8, mask_batch, light_pt, B, L, H, W, N, t_table, bonus, bonus_box,
                           min_dist, argmin, workspace, workspace_bytes, stream, FusedShade{}, opt);
}

extern "C" int gcfr_render_fwd(const float *light_raw, int32_t clamp_z, float clamp_min,
                               float light_distance, const float *depth, const uint8_t *mask_u8,
                               int32_t mask_batch, const float *normals, const float *albedo,
                               const float *ambient, int32_t B, int32_t L, int32_t H, int32_t W,
                               int32_t N, const double *t_table, float bonus, const float *bonus_box,
                               float intensity, float *unit_out, float *light_pt_out, float *min_dist,
                               int32_t *argmin, float *shadow_w, float *full, float *final_shading,
                               float *rendered, void *workspace, size_t workspace_bytes, void *stream,
                               const gcfr_options *opt)
{
    const bool pre_only = prepass_only(opt);  // (gcfr_options.phase: the prepass reads depth, mask, light_raw and t_table only)
    if (!light_raw || !unit_out || !light_pt_out || !workspace || (!pre_only && (!normals || !albedo || !ambient || !rendered)))
        return GCFR_ERR_INVALID_ARGUMENT;
    if (B <= 0 || L <= 0)
        return GCFR_ERR_INVALID_ARGUMENT;
    FusedShade fs;
    fs.lights.light_raw = light_raw;  // light prep rides in the prepass launch
    fs.lights.unit_out = unit_out;
    fs.lights.light_pt_out = light_pt_out;
    fs.lights.L = L;
    fs.lights.clamp_z = clamp_z;
    fs.lights.clamp_min = clamp_min;
    fs.lights.light_distance = light_distance;
    fs.normals = normals;
    fs.albedo = albedo;
    fs.ambient = ambient;
    fs.shadow_w = shadow_w;
    fs.full = full;
    fs.final_shading = final_shading;
    fs.rendered = rendered;
    fs.intensity = intensity;
    return shadow_fwd_impl(depth, mask_u8, mask_batch, light_pt_out, B, L, H, W, N, t_table, bonus,
                           bonus_box, min_dist, argmin, workspace, workspace_bytes, stream, fs, opt);
}

extern "C" int gcfr_render_from_depth_fwd(const float *light_raw, int32_t clamp_z, float clamp_min,
                                          float light_distance, const float *depth,
                                          const uint8_t *mask_u8, int32_t mask_batch, double fx, double fy,
                                          double cx, double cy, float z_offset, int32_t negate_y,
                                          const float *albedo, const float *ambient, int32_t B, int32_t L,
                                          int32_t H, int32_t W, int32_t N, const double *t_table,
                                          float bonus, const float *bonus_box, float intensity,
                                          float *unit_out, float *light_pt_out, float *min_dist,
                                          int32_t *argmin, float *normals_out, float *shadow_w, float *full,
                                          float *final_shading, float *rendered, void *workspace,
                                          size_t workspace_bytes, void *stream, const gcfr_options *opt)
{
    const bool pre_only = prepass_only(opt);
    if (!light_raw || !unit_out || !light_pt_out || !workspace || (!pre_only && (!albedo || !ambient || !rendered)) ||
        B <= 0 || L <= 0 || fx == 0.0 || fy == 0.0)
        return GCFR_ERR_INVALID_ARGUMENT;
    FusedShade fs;
    fs.lights.light_raw = light_raw;
    fs.lights.unit_out = unit_out;
    fs.lights.light_pt_out = light_pt_out;
    fs.lights.L = L;
    fs.lights.clamp_z = clamp_z;
    fs.lights.clamp_min = clamp_min;
    fs.lights.light_distance = light_distance;
    fs.normals = nullptr;  // computed in the epilogue
    set_focal(fs.nrm, fx, fy);
    fs.nrm.cx = cx;
    fs.nrm.cy = cy;
    fs.nrm.z_offset = z_offset;
    fs.nrm.negate_y = negate_y;
    fs.normals_out = normals_out;
    fs.albedo = albedo;
    fs.ambient = ambient;
    fs.shadow_w = shadow_w;
    fs.full = full;
    fs.final_shading = final_shading;
    fs.rendered = rendered;
    fs.intensity = intensity;
    return shadow_fwd_impl(depth, mask_u8, mask_batch, light_pt_out, B, L, H, W, N, t_table, bonus,
                           bonus_box, min_dist, argmin, workspace, workspace_bytes, stream, fs, opt);
}

// ----------------------------------------------------------------------------------------------
// measurement aid: the achievable-HBM probe (bench.py)
// ----------------------------------------------------------------------------------------------
namespace gcfr {
// One workgroup per CU, four 16-byte loads in flight per lane, non-temporal both ways: the best of tools/copy_probe_sweep.hip's
// sweep over grid size / loads in flight / cache policy (6.27 TB/s read + write on 1 GiB; 8,192 workgroups of plain float4: 4.7).
__global__ __launch_bounds__(256) void copy_probe_kernel(const f32x4 *__restrict__ src, f32x4 *__restrict__ dst, size_t n)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n; i += 4 * stride) {
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            v[u] = __builtin_nontemporal_load(src + i + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            __builtin_nontemporal_store(v[u], dst + i + u * stride);
    }
    for (; i < n; i += stride)
        dst[i] = src[i];
}
}  // namespace gcfr

extern "C" int gcfr_copy_probe(const void *src, void *dst, size_t bytes, void *stream)
{
    if (!src || !dst || bytes == 0 || (bytes & 15u) || ((uintptr_t)src & 15u) || ((uintptr_t)dst & 15u))
        return GCFR_ERR_INVALID_ARGUMENT;
    hipLaunchKernelGGL(gcfr::copy_probe_kernel, dim3(256), dim3(256), 0, (hipStream_t)stream, (const gcfr::f32x4 *)src, (gcfr::f32x4 *)dst,
                       bytes / 16);
    return launch_status();
}
