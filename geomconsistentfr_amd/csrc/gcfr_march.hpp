// The march of the ray-traced shadow (gfx950): everything the sample loop is made of -- shared device helpers, the tile
// function march_tile(), the kernels built from it and the launcher templates.  Header-only: the kernels are templates over
// (tile width, samples per group, ...) and each (tile width, group) pair is instantiated in its own translation unit
// (gcfr_march_unit.hip, compiled once per pair by build.py with -DGCFR_UNIT_TILE_W / -DGCFR_UNIT_GROUP, in parallel), which
// gcfr_shadow.hip -- prepass kernels, plain march kernel, C ABI -- calls through launch_march_unit().  One translation unit
// with all 200 kernels took 95 s to compile.
//
// Replaces train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:371-515 (and, fused, :364-369 / :517-522); see gcfr_shadow.hip
// for the overview and include/gcfr.h for the entry points.
#pragma once

#include "gcfr_device.hpp"
#include "gcfr_mutants.hpp"

#include "../../include/gcfr.h"

#include <atomic>
#include <cstddef>
#include <type_traits>

namespace gcfr {

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
    double x64, y64, dx64, dy64, halfW, halfH;
    int H, W;
};

// census: tie re-march (plain sample)
__device__ inline float ray_sample(const RayConst &rc, double t, __amdgpu_buffer_rsrc_t zr,
                                   __amdgpu_buffer_rsrc_t mr, bool &masked)
{
    const int W = rc.W, H = rc.H;
    const double sx = rc.x64 + t * rc.dx64;  // T8:472 / 480 (f64, mul and add rounded separately)
    const double sy = rc.y64 + t * rc.dy64;
    // rounded cell -> mask lookup (T8:472-477, 510)
    const int col_r = (int)(__builtin_rint(sx) + rc.halfW);
    const int row_r = (int)(rc.halfH - __builtin_rint(sy));
    // unrounded position (T8:480-487)
    const double ux = (sx + rc.halfW) - 0.0001;
    const double uy = (rc.halfH - sy) - 0.0001;
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
    const float Ax = (float)(ux - rc.halfW), Ay = (float)(rc.halfH - uy), Az = (float)zA;
    const float BAx = Ax - rc.x, BAy = Ay - rc.y, BAz = Az - rc.zb;
    const float Xx = __builtin_fmaf(BAy, rc.BCz, -(BAz * rc.BCy));  // torch.cross uses fma
    const float Xy = __builtin_fmaf(BAz, rc.BCx, -(BAx * rc.BCz));
    const float Xz = __builtin_fmaf(BAx, rc.BCy, -(BAy * rc.BCx));
    masked = (mk == 0);
    return ((Xx * Xx + Xy * Xy) + Xz * Xz) + kEps4;
}

// torch.min (T8:514) returns the FIRST index of the minimal DISTANCE d = sqrt(S)/|BC|.  sqrt and the division
// are monotone, so the minimal distance is the distance of the minimal S, but several slightly larger S (up to
// about nine consecutive floats) round to the same distance.  Every sample of that tie class that precedes
// the minimum is a running minimum when it is met, so the class is a suffix of the chain of running minima
// and the march tracks the chain's last link: if the predecessor does not tie nothing does.  If it does, an
// even earlier link may tie as well; this re-marches [0, prevk) for the (rare) lanes concerned and returns
// the first unmasked sample whose distance equals d.  Wave-uniform loop, per-lane predicate.
__device__ inline int first_tied_sample(const RayConst &rc, const double *t_table, __amdgpu_buffer_rsrc_t zr,
                                        __amdgpu_buffer_rsrc_t mr, bool tie, int prevk, float den, float d)
{
    int first = prevk;
    int k_hi = tie ? prevk : 0;
    // wave maximum (6 DPP-free steps are fine here: rare path)
    for (int off = 32; off > 0; off >>= 1)
        k_hi = max(k_hi, __shfl_xor(k_hi, off));
    k_hi = __builtin_amdgcn_readfirstlane(k_hi);
    bool found = !tie;
    for (int k = 0; k < k_hi; ++k) {
        bool masked;
        const float S = ray_sample(rc, t_table[k], zr, mr, masked);
        const bool hit = !found && (k < prevk) && !masked && (__builtin_sqrtf(S) / den == d);
        first = hit ? k : first;
        found = found || hit;
    }
    return first;
}



// ----------------------------------------------------------------------------------------------
// shadow march, "quad texel" variant (used when the caller provides a workspace)
//
// A prepass rewrites each depth map as a (H+1) x (W+1) grid of 2x2 neighbourhoods
//     Q[r][c] = { z[r][c], z[r][c+1], z[r+1][c], z[r+1][c+1] },   r in [-1, H-1], c in [-1, W-1],
// with row/column -1 holding the reference's wrap-around neighbours (index -1 == last, T8:488-491).
// The march then needs ONE 16-byte gather and one address per ray-step instead of four 4-byte
// gathers, four addresses and the wrap arithmetic; values are the same bits, so results are
// bit-identical to the direct kernel above (tests/test_gpu_parity.py asserts it).
// When floor(u) == ceil(u) (u integral) the reference reads z[f] twice with weights 0 and 0; here the
// second operand is z[f+1], still multiplied by 0 -- identical for finite depth.
//
// Other exact instruction trims in this variant:
//   * rint(s) via the 2^52+2^51 magic add (round-half-even of the f64 adder == torch.round), the
//     integer falls out of the low dword with no v_rndne / v_cvt; when W/2 and H/2 are even the
//     +W/2 and H/2- offsets ride in the magic constant (parity-safe), otherwise they are int adds;
//   * argmin tracking compiled out when the caller does not ask for it (inference).
// ----------------------------------------------------------------------------------------------
struct PrepassLights {  // optional: fold gcfr_light_prep into the prepass launch (gcfr_render_fwd)
    const float *light_raw = nullptr;
    float *unit_out = nullptr, *light_pt_out = nullptr;
    int L = 0, clamp_z = 0;
    float clamp_min = 0.0f, light_distance = 0.0f;
};

constexpr int kBBoxInit = 0x7f7f7f7f;  // "+infinity" for the int minima below

// Wave-wide integer minimum, result wave-uniform (SGPR).  DPP row shifts + row broadcasts (gfx9): six
// VALU-speed steps instead of six dependent ds_bpermute round trips (__shfl_xor) in the kernels'
// latency-bound prologues (measured: fixed cost of the march at B=8 20.5 -> 19.5 us; LDS atomics instead
// were 2.4x worse).
// census: wave reductions (DPP)
template <int CTRL, int ROW_MASK>
__device__ inline int dpp_min_step(int v)
{
    const int moved = __builtin_amdgcn_update_dpp(0x7fffffff, v, CTRL, ROW_MASK, 0xf, false);
    return min(v, moved);
}
__device__ inline int wave_min_i32(int v)
{
    v = dpp_min_step<0x111, 0xf>(v);  // row_shr:1
    v = dpp_min_step<0x112, 0xf>(v);  // row_shr:2
    v = dpp_min_step<0x114, 0xf>(v);  // row_shr:4
    v = dpp_min_step<0x118, 0xf>(v);  // row_shr:8   -> lane 15 of each 16-lane row holds the row minimum
    v = dpp_min_step<0x142, 0xa>(v);  // row_bcast:15 into rows 1 and 3
    v = dpp_min_step<0x143, 0xc>(v);  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave minimum
    return __builtin_amdgcn_readlane(v, 63);
}

// float <-> int with the same ordering (so the integer DPP minimum above serves floats too)
__device__ inline int f32_sortable(float f)
{
    const int i = __builtin_bit_cast(int, f);
    return i ^ ((i >> 31) & 0x7fffffff);
}
__device__ inline float f32_unsortable(int i)
{
    return __builtin_bit_cast(float, i ^ ((i >> 31) & 0x7fffffff));
}

// ----------------------------------------------------------------------------------------------
// Depth bounds ("hierarchical z") grid.  Tile (i, j) of stride s = 2^ls bounds the depth (a band around a plane,
// see build_zbounds_block) over the 2s x 2s cells whose EXTENDED indices (row r+1, column c+1, with r = c = -1 the
// reference's wrap-around to the last row / column) lie in [i*s, i*s + 2s) x [j*s, j*s + 2s): tiles
// overlap by half, so any footprint of at most s+1 cells per axis lies inside the tile that its lowest
// index selects.  The march uses it to skip sample groups that provably cannot lower a lane's running
// minimum (see shadow_fwd_quad_kernel); s is the smallest power of two >= 8 that covers the cells one
// group of `group` consecutive samples can touch, derived from the sample table on the device by both
// kernels.  The host sizes the grid for s = 8.
// ----------------------------------------------------------------------------------------------
// census: bounds grid stride
template <class TablePtrT>
__device__ inline int zb_log2_stride(int H, int W, int N, TablePtrT t_table, int group, bool *fits = nullptr)
{
    if (fits)
        *fits = false;
    if (N < 2)
        return 3;
    const float step = fabsf((float)((t_table[N - 1] - t_table[0]) / (double)(N - 1)));
    // rint(s) moves by <= floor(fd) + 1 cells over a group (1.002: the table may deviate 0.1 % from uniform, see
    // the prepass' table check, plus the f32 roundings here)
    const float fd = (float)(group - 1) * step * (float)max(H, W) * 1.002f;
    const int need = (int)fminf(fmaxf(fd, 0.0f), 1024.0f) + GCFR_M(22, 1, 3);      // + the cell either side (floor / ceil)
    int ls = 3;
    while ((1 << ls) < need && ls < 5)  // capped at 32: coarser tiles bound nothing (their footprints fail the coverage test)
        ++ls;
    if (fits)
        *fits = (1 << ls) >= need;  // every group's footprint lies in the tile its lowest cell selects
    return ls;
}

// records per image: the tiles at the finest stride plus one sentinel (-inf, +inf) that uncovered footprints read
__host__ __device__ inline int zb_max_tiles(int H, int W) { return ((H >> 3) + 1) * ((W >> 3) + 1) + 1; }
// per-image stride of the records, a whole number of 1-KiB pieces: the march's LDS-staged variant copies an image's
// records with global_load_lds_dwordx4, 64 lanes x 16 B per instruction
__host__ __device__ inline int zb_stride(int H, int W) { return (zb_max_tiles(H, W) + 63) & ~63; }
// mask bitmap of the LDS-staged march: one bit per cell (1 = mask cell non-zero), row-major, 32 cells per dword,
// per-mask stride padded to 1 KiB; needs W % 32 == 0
__host__ __device__ inline int bitmap_stride_bytes(int H, int W) { return (((H * W) >> 3) + 1023) & ~1023; }
// "horizon" tables of the trailing loop's termination test (build_horizon_block): per image four arrays of running maxima
// {col_pre, col_suf, row_pre, row_suf} of the depth an unmasked sample can read, each kHorizonDim entries and CENTRED: entry
// kHorizonDim/2 + X belongs to image-plane column X = c - W/2 (row entry kHorizonDim/2 - Y to Y = H/2 - r), entries outside the
// image repeat the nearest one -- so the march indexes them with compile-time constants only (its scalar registers are all
// taken).  An entry is a float4: the values of the kHorizonBands row bands the prepass builds the tables from; the value proper
// is their maximum.  The tables sit BEHIND the image's depth-bounds records, inside the same per-image slot, and are reached
// through the buffer descriptor the march already holds for the records.
constexpr int kHorizonDim = 1024;
constexpr int kHorizonBands = 4;
__host__ __device__ inline bool hz_shape_ok(int H, int W) { return ((W & 3) == 0) && W <= kHorizonDim && H <= kHorizonDim; }
// per-image slot of [records | horizon tables], in records (float4), a whole number of 1-KiB pieces
__host__ __device__ inline int zb_slot(int H, int W) { return hz_shape_ok(H, W) ? zb_stride(H, W) + 4 * kHorizonDim : zb_stride(H, W); }

// Sum over each 16-lane row of the wave, result in every lane of the row's last lane ... read with readlane(row*16+15).
// census: wave reductions (DPP)
__device__ inline float row_sum_f32(float v)
{
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, false));
    return v;  // lane 15 of each row holds the row's sum
}
__device__ inline float lane_value(float v, int lane)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// One wave per tile; `block` counts the 4-wave workgroups assigned to this job (the head of the prepass grid).
// Record {a, b, c_lo, c_hi}: every cell of the tile satisfies  a X + b Y + c_lo <= z <= a X + b Y + c_hi  with
// (X, Y) the cell's coordinates in the kernel's frame (X = column - W/2, Y = H/2 - row; the wrap row / column sit
// at row / column -1).  A bilinear sample is a convex combination of four cells whose weighted mean position is
// the sample position, so the same band bounds it AT the sample position -- on a smooth surface the band is
// curvature-sized where a plain min/max is slope-sized.  (a, b) comes from the means of the tile's four
// quadrants, clamped to +-4; any (a, b) is valid, the residual extrema make it so.

// Per-image statistics the march needs before it starts: the bounding box of the mask's non-zero cells and the
// depth range.  One 256-thread block per chunk of kStatChunk pixels writes ONE partial record (four + two minima:
// {r_min, c_min, -r_max, -c_max}, {z_min, -z_max} as sortable ints; kBBoxInit / INT_MAX where the chunk has
// nothing to report), so an image has only P/16384 partials (4 at 256x256, 8 at 512x512) and every march wave reduces them
// itself with one load and six DPP minima -- no atomics to initialise, no workgroup barrier in the march (round 1
// kept 256 partials per image and reduced them through LDS in every march workgroup's prologue).
constexpr int kQueueSlot = 64;  // (workspace layout: the table record keeps a 256-B line to itself)
enum { kTfOk = 0, kTfStride = 1, kTfTabs = 2, kTfTfirst = 3, kTfInvDt = 4 };  // tflag[]: per-launch facts about the sample table (prepass)
constexpr int kStatChunk = 16384;
// Pixels per statistics chunk: kStatChunk -- doubled for images of 9 ... 16 plain chunks (up to 512 x 512, BASELINE
// configs[4]'s size), so that those too have at most 8 records and every march wave folds them on the scalar unit
// (reduce_image_stats; sixteen records there cost the march its registers: the scalar loads spill into VGPRs).
__host__ __device__ inline int stat_chunk_shift(int H, int W)
{
    const int raw = (H * W + kStatChunk - 1) >> 14;
    return (raw > 8 && raw <= 16) ? 15 : 14;
}
__host__ __device__ inline int stat_chunk_px(int H, int W) { return 1 << stat_chunk_shift(H, W); }
__host__ __device__ inline int n_stat_chunks(int H, int W)
{
    const int sh = stat_chunk_shift(H, W);
    return (H * W + (1 << sh) - 1) >> sh;
}
static_assert(kStatChunk == 1 << 14, "stat_chunk_shift() spells kStatChunk as a shift");


// Operands of the per-pixel epilogue (distance finish, optional fused shading).  They live in the kernel-argument
// segment like the rest of ShadowQuadArgs, but the march reads them through an opaque pointer AFTER the sample loop
// (epilogue_args()): referenced through the by-value struct the compiler loads every pointer at kernel entry and
// keeps ~40 SGPRs live across the loop, which pushed the loop's wave-uniform f64 constants into VGPRs (round 1:
// 106 SGPRs, 28 B/lane of scratch at the forced occupancy).
struct MarchEpilogueArgs {
    float *min_dist;        // (B,L,H,W)
    int32_t *argmin;        // (B,L,H,W) or null
    float bonus, bx_lo, bx_hi, by_lo, by_hi;
    // fused shading epilogue (FUSE_SHADE): T8:364-369, 517-522 on the pixel the lane just marched
    const float *normals;   // (B,3,H,W); nullptr: the epilogue computes the normal from the depth stencil (T8:353-354)
    const float *albedo;    // (B,3,H,W)
    const float *ambient;   // (B,L)
    float *shadow_w, *full, *final_shading, *rendered;
    float intensity;
    NormalsArgs nrm;
    float *normals_out;     // (B,3,H,W) or null
};

struct ShadowQuadArgs {
    const float *depth;     // (B,H,W)      own-pixel depth
    const float4 *quad;     // (B,H+1,W+1)  prepass output
    const int *bbox;        // (MB,n_stat,4) prepass output: partial mask bounding boxes {r_min, c_min, -r_max, -c_max}
    const int *diag;        // (MB,n_stat,4) prepass output: partial diagonal extents {min(c+r), -max(c+r), min(c-r), -max(c-r)}
    const float4 *zb;       // (B,zb_slot) prepass output: depth bounds grid {a, b, c_lo, c_hi}, or null (skip off)
    const int *zrange;      // (B,n_stat,2) prepass output: partial depth ranges {z_min, -z_max} (sortable ints)
    const int *mones;       // (MB,n_stat)  prepass output: 1 iff every mask cell of the chunk is non-zero
    int *tflag;             // [0] prepass output: 1 iff the sample table is increasing, inside [0,1] and uniform
    const uint8_t *mask;    // (MB,H,W)
    const uint32_t *bitmap; // (MB, bitmap_stride_bytes / 4) prepass output (LDS-staged march): one bit per mask cell
    const float *light_pt;  // (B,L,3)
    const double *t_table;  // (N)
    unsigned long long *counters;  // GCFR_COUNTERS builds: work counts, see gcfr_options
    int32_t mask_batch, B, L, H, W, N;
    int32_t tiles_x, tiles_y;  // tiles per image row / column
    int32_t bl_offset;         // first (image, light) index of this launch (grid z is limited to 65535)
    int32_t hz_off;            // byte offset of the horizon tables behind each image's depth-bounds records; -1: not built
    MarchEpilogueArgs epi;
};

// The march kernels take ShadowQuadArgs by value as their only argument and read it IN PLACE from the
// kernel-argument segment through this pointer (scalar loads), not through the by-value copy: referenced by value
// the compiler loads every field at kernel entry and keeps it in SGPRs for the kernel's lifetime -- ~40 SGPRs of
// epilogue pointers live across the sample loop, and in the persistent schedule everything live across the tile loop
// (round 1: 106 SGPRs, the loop's wave-uniform f64 constants pushed into VGPRs, 28 B/lane of scratch).  An opaque
// redefinition of the pointer (`launder`) at the top of each tile and before the epilogue makes the loads after
// it un-hoistable, so each phase keeps only its own operands.
typedef const __attribute__((address_space(4))) ShadowQuadArgs *ArgPtr;
typedef const __attribute__((address_space(4))) MarchEpilogueArgs *EpiPtr;
__device__ __forceinline__ ArgPtr kernel_args()
{
    return (ArgPtr)__builtin_amdgcn_kernarg_segment_ptr();  // the struct starts the segment
}
template <class T>
__device__ __forceinline__ T launder(T p)
{
    asm volatile("" : "+s"(p));
    return p;
}

constexpr double kRintMagic = 6755399441055744.0;  // 2^52 + 2^51
typedef float f32x4 __attribute__((ext_vector_type(4)));

// census: rint magic (lo32)
__device__ inline int lo32(double v)
{
    return (int)(unsigned)(__builtin_bit_cast(unsigned long long, v) & 0xffffffffull);
}

// Work counters of the counting build (-DGCFR_COUNTERS; tools/count_work.py): wave-uniform tallies, added to
// gcfr_options.counters once per tile.  Compiled out of the product build.
enum { kCntTiles, kCntGroupsNominal, kCntGroupsVisited, kCntBoundTests, kCntBodies, kCntLaneSamples, kCntEarlyExit,
       kCntTieRemarch, kCntSamplesInRange, kCntBoundsGivenUp, kCntVisitsAfterLastBody, kCntVisitsBeforeFirstBody,
       kCntTrailEnter, kCntTrailSkips, kCntTrailLeave, kCntRoughSamples, kCntWaveSamples, kCntWaveSamplesTaken, kCntLaneTakes,
       // (-DGCFR_AUDIT, see below) lane-samples a claim spoke for / samples that contradict it; the last one is a maximum, not a sum
       kCntAuditBoundChecks, kCntAuditBoundViol, kCntAuditTermChecks, kCntAuditTermViol, kCntAuditMaskedChecks, kCntAuditMaskedViol,
       kCntAuditSafeViol, kCntAuditMaxUse, kCntUsed };
static_assert(kCntUsed <= GCFR_N_COUNTERS, "gcfr_options.counters holds GCFR_N_COUNTERS tallies");
// The AUDIT build (-DGCFR_COUNTERS -DGCFR_AUDIT; tools/audit.py, tests/test_gpu_audit.py; round 5).  Every claim the march makes
// about samples it does NOT evaluate is checked against the plain evaluation of exactly those samples (ray_sample(): the depth plane
// and the mask, no workspace, no bounds), whether or not the claim was decisive for a result:
//   * the depth-bound test: "g > 0  =>  S_k >= 0.998 g^2 for every unmasked sample k of the group" -- at EVERY evaluation of the
//     bound, not only where it skipped something (kCntAuditBound*); and how much of the error budget Kerr the evaluation used up,
//     (raw bound - sqrt(S_k)) / Kerr, as a maximum in 1/1000 (kCntAuditMaxUse: 1000 would be a bound without any margin left);
//   * early termination, main and trailing loop: the same for every later sample of a lane that is finished by the bound (kCntAuditTerm*);
//   * "this sample is masked": the candidate range (every sample outside [lane_lo, lane_hi]), lane_last at terminations and in the
//     trailing loop's skips (kCntAuditMasked*);
//   * "this lane's distance is certainly below the masked value" wherever bestS < safeS lets a lane ignore its any_masked (kCntAuditSafeViol).
// An end-to-end comparison only sees a wrong claim when it changes a minimum; the audit sees it whenever it is made -- which is what
// kills the mutants whose margin is shadowed by its neighbours in every decisive case (csrc/gcfr_mutants.hpp: 3 and 5).
#if defined(GCFR_AUDIT) && !defined(GCFR_COUNTERS)
#error "-DGCFR_AUDIT needs -DGCFR_COUNTERS"
#endif
#ifdef GCFR_COUNTERS
#define GCFR_COUNT(i, n) (cnt[i] += (unsigned)(n))
#else
#define GCFR_COUNT(i, n) ((void)0)
#endif

// Per-image statistics, wave-uniform: the reduction of the prepass' partial records (build_stats_block).
// census: image statistics (reduce_image_stats)
struct ImageStats {
    int r_min, c_min, r_max, c_max;  // bounding box of the mask's non-zero cells (r_min == kBBoxInit: none)
    int s_min, s_max, d_min, d_max;  // ... and their diagonal extents: c + r and c - r (the bounding octagon)
    int gz_lo_s, gz_nhi_s;           // depth range {z_min, -z_max} as sortable ints
    int mask_all_ones;               // 1 iff the image's mask has no zero cell at all
};
__device__ inline ImageStats reduce_image_stats(ArgPtr a, int b, int lane, bool want_z)
{
    const int n = n_stat_chunks(a->H, a->W);
    if (n <= 8) {
        // Up to 512 x 256 pixels: the handful of chunk records is folded on the SCALAR unit (constant-address-space loads of
        // uniform addresses are s_loads, the minima s_min_i32) -- the vector form below costs every tile ~100 VALU
        // instructions (seven 64-lane DPP reductions), 4 % of the march's instruction count at B=8 x 256^2.
        typedef const __attribute__((address_space(4))) int *ConstI32Ptr;
        const ConstI32Ptr sb = (ConstI32Ptr)(unsigned long long)a->bbox + 4 * (size_t)(a->mask_batch == 1 ? 0 : b) * n;
        const ConstI32Ptr sz = (ConstI32Ptr)(unsigned long long)a->zrange + 2 * (size_t)b * n;
        const ConstI32Ptr so = (ConstI32Ptr)(unsigned long long)a->mones + (size_t)(a->mask_batch == 1 ? 0 : b) * n;
        const ConstI32Ptr sd = (ConstI32Ptr)(unsigned long long)a->diag + 4 * (size_t)(a->mask_batch == 1 ? 0 : b) * n;
        int d0 = kBBoxInit, d1 = kBBoxInit, d2 = kBBoxInit, d3 = kBBoxInit;
        int m0 = kBBoxInit, m1 = kBBoxInit, m2 = kBBoxInit, m3 = kBBoxInit, z0 = 0x7fffffff, z1 = 0x7fffffff, ones = 1;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (j < n) {
                m0 = min(m0, sb[4 * j + 0]);
                m1 = min(m1, sb[4 * j + 1]);
                m2 = min(m2, sb[4 * j + 2]);
                m3 = min(m3, sb[4 * j + 3]);
                ones = min(ones, so[j]);
                d0 = min(d0, sd[4 * j + 0]);
                d1 = min(d1, sd[4 * j + 1]);
                d2 = min(d2, sd[4 * j + 2]);
                d3 = min(d3, sd[4 * j + 3]);
                if (want_z) {
                    z0 = min(z0, sz[2 * j + 0]);
                    z1 = min(z1, sz[2 * j + 1]);
                }
            }
        }
        ImageStats st;
        st.r_min = m0;
        st.c_min = m1;
        st.r_max = -m2;
        st.c_max = -m3;
        st.gz_lo_s = z0;
        st.gz_nhi_s = z1;
        st.mask_all_ones = ones;
        st.s_min = d0;
        st.s_max = -d1;
        st.d_min = d2;
        st.d_max = -d3;
        return st;
    }
    const int4 *pb = (const int4 *)a->bbox + (size_t)(a->mask_batch == 1 ? 0 : b) * n;
    const int2 *pz = (const int2 *)a->zrange + (size_t)b * n;
    const int *po = a->mones + (size_t)(a->mask_batch == 1 ? 0 : b) * n;
    const int4 *pd = (const int4 *)a->diag + (size_t)(a->mask_batch == 1 ? 0 : b) * n;
    int4 m = make_int4(kBBoxInit, kBBoxInit, kBBoxInit, kBBoxInit), dg = m;
    int2 mz = make_int2(0x7fffffff, 0x7fffffff);
    int ones = 1;
    for (int j = lane; j < n; j += 64) {
        ones = min(ones, po[j]);
        const int4 v = pb[j];
        m.x = min(m.x, v.x);
        m.y = min(m.y, v.y);
        m.z = min(m.z, v.z);
        m.w = min(m.w, v.w);
        const int4 w = pd[j];
        dg.x = min(dg.x, w.x);
        dg.y = min(dg.y, w.y);
        dg.z = min(dg.z, w.z);
        dg.w = min(dg.w, w.w);
        if (want_z) {
            const int2 vz = pz[j];
            mz.x = min(mz.x, vz.x);
            mz.y = min(mz.y, vz.y);
        }
    }
    ImageStats st;
    st.r_min = wave_min_i32(m.x);
    st.c_min = wave_min_i32(m.y);
    st.r_max = -wave_min_i32(m.z);
    st.c_max = -wave_min_i32(m.w);
    st.gz_lo_s = want_z ? wave_min_i32(mz.x) : 0x7fffffff;
    st.gz_nhi_s = want_z ? wave_min_i32(mz.y) : 0x7fffffff;
    st.mask_all_ones = wave_min_i32(ones);
    st.s_min = wave_min_i32(dg.x);
    st.s_max = -wave_min_i32(dg.y);
    st.d_min = wave_min_i32(dg.z);
    st.d_max = -wave_min_i32(dg.w);
    return st;
}

// One tile of one (image, light) pair: 64 lanes = 64 pixels.
// SPLIT = 0: one wave marches all N samples of the tile; the waves of a workgroup march different tiles and never
//            synchronise.
// SPLIT = 1: the 4 waves of the workgroup march the SAME tile, a contiguous quarter of the sample range each
//            (four gathers in flight per body, natural occupancy), and combine their partial minima through LDS
//            (earliest index wins ties, as torch.min): tiny, latency-bound launches (one or two images).
// (Round 2 also built and measured four more schedules on this tile function -- persistent waves with a tile queue or a
//  strided assignment, four cooperating waves per tile, work stealing inside the workgroup, helping across the chip;
//  all bit-identical, all slower: profiles/r02_schedule_experiments.md.  Their code was removed from the product source
//  in round 3; it builds from commit 4db51f3 with -DGCFR_EXPERIMENTAL_SCHEDULES.  Round 3 added the one round 2 had left
//  untried -- a budgeted first pass plus a second launch that resumes the unfinished tiles four ways from warm minima --
//  measured it (bit-identical, 79-87 us against 68: profiles/r03_twopass_ab.md) and took it out again: commit 7e0eaa4.)
#ifndef GCFR_TILE_INLINE
#define GCFR_TILE_INLINE __forceinline__
#endif

// LDS image of the LDS-staged march (dynamic shared memory, sized by the launch): [mask bitmap, bitmap_stride_bytes |
// depth-bounds records, zb_stride * 16 B] of the workgroup's image.  26 KiB at 256 x 256: six workgroups per CU, which is
// what the march's forced occupancy (six waves per SIMD, four waves per workgroup) needs.
extern __shared__ uint4 gcfr_lds_stage[];

// Copy the image's bitmap (unless its mask has no zero cell: then the march never reads it) and bounds records into LDS:
// 1-KiB pieces, piece i by wave i mod 4, each ONE global_load_lds_dwordx4 -- global -> LDS without passing through
// registers, issued at kernel entry and awaited by the workgroup barrier in front of the sample loop, so the copy runs
// behind the tile prologue's ~750 instructions.
// census: LDS staging
__device__ inline void stage_lds(ArgPtr a, int b, bool with_bitmap)
{
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int H = a->H, W = a->W;
    const int bm_bytes = bitmap_stride_bytes(H, W), zb_bytes = a->zb ? zb_stride(H, W) * 16 : 0;
    const char *gbm = (const char *)a->bitmap + (size_t)(a->mask_batch == 1 ? 0 : b) * bm_bytes;
    const char *gzb = (const char *)a->zb + (size_t)b * zb_slot(H, W) * 16;
    const int n_bm = with_bitmap ? (bm_bytes >> 10) : 0, n_zb = zb_bytes >> 10;
    for (int ch = wave; ch < n_bm + n_zb; ch += 4) {
        const bool is_bm = ch < n_bm;
        const int piece = is_bm ? ch : ch - n_bm;
        const char *src = (is_bm ? gbm : gzb) + ((size_t)piece << 10) + (lane << 4);
        const int dst = (is_bm ? 0 : bm_bytes) + (piece << 10);  // wave-uniform LDS byte offset; lane i lands at + 16 i
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)((__attribute__((address_space(3))) char *)gcfr_lds_stage + dst),
                                         16, 0, 0);
    }
}
// lane id from the hardware (two VALU), opaque to the optimiser: a value derived from it has no live range before this point
// census: tile set-up: lane, pixel, descriptors, light, own depth
__device__ inline int fresh_lane_id()
{
    int l;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
    return l;
}

// OWN (gcfr_options.pixels = 1, round 4): pixels whose OWN mask cell is zero are not marched -- their lanes contribute no
// sample range, never ask for a body, and get the masked value (minimum distance 1e6, argmin -1, hence shadow weight 1);
// a tile without an unmasked pixel does no march at all.  The one knob that changes results, and only at those pixels:
// every consumer of the reference multiplies them by the mask (T8:619, 633, 641, 643).  See include/gcfr.h.
// MODE (round 4): what a tile does when it marches WITHOUT the depth bounds (it gave them up -- see the give-up test --, the
// caller switched them off, or the sample table is not one they reason about).
//   kModeInline  the loops below run with `use_zb` false (k-split and LDS-staged kernels);
//   kModeFull    the tile RETURNS TRUE instead, before anything is marched or written, and the caller runs it again as
//   kModeRough   the rough loop only: no bounds machinery compiled in at all.
// The grid kernels hold both as sibling regions (march_grid): the rough loop inside the bounds variant, as a branch behind
// its prologue, made the register allocator re-plan the main loops -- the bounds record of the group in flight went to
// scratch in the six-wave kernel, reloaded at every test -- whereas separate instantiations do not interfere.  The price
// is a second prologue (~5 % of a rough tile's work).
enum { kModeInline = 0, kModeFull = 1, kModeRough = 2 };
#ifndef GCFR_ROUGH
#define GCFR_ROUGH 1
#endif
#ifndef GCFR_ROUGH_CHUNK
#define GCFR_ROUGH_CHUNK 3      // samples in flight per iteration of the rough loop: six-wave inference march (round 4: 2 -- 3 spilled
                                // then; since round 5's trims 3 and 4 fit the 80 registers: noise-400 faces +2.2 % / +0.7 %, the bench
                                // faces, the FFHQ fixtures and all-ones masks unchanged, profiles/r05_rough_chunk_ab.txt)
#endif
#ifndef GCFR_ROUGH_CHUNK_ARGMIN
#define GCFR_ROUGH_CHUNK_ARGMIN 3   // ... five-wave training march
#endif
template <int TILE_W, bool EVEN_HALF, bool WANT_ARGMIN, int DEPTH, bool FUSE_SHADE, int SPLIT, bool ALL_ONES = false, bool LDS = false,
          bool OWN = false, int MODE = kModeInline>
__device__ GCFR_TILE_INLINE bool march_tile(ArgPtr a, const int bl, const int qy, const int tx,
                                           const ImageStats &st)
{
    static_assert(MODE == kModeInline || (SPLIT == 0 && !LDS), "full / rough pairs: the grid schedule's global-memory kernels");
    constexpr int TILE_H = 64 / TILE_W;
    static_assert(SPLIT == 0 || SPLIT == 1, "SPLIT: 0 = one wave per tile, 1 = the workgroup's four waves split the sample range");
    static_assert(!(LDS && SPLIT != 0), "the LDS-staged march is a throughput variant: one wave per tile");
    static_assert(!(OWN && (LDS || SPLIT != 0 || ALL_ONES)), "pixels = mask: the grid schedule's global-memory variant only");
// census: tile set-up: lane, pixel, descriptors, light, own depth
    const int H = a->H, W = a->W, L = a->L;
    // Wave-uniform read-only inputs are read through the CONSTANT address space: in the persistent schedule the
    // previous tile's stores and the queue atomic precede these loads in program order, so through a plain global
    // pointer the compiler can no longer prove the memory unclobbered and turns every sample-table read of the
    // sample loop into a VECTOR load (measured: the persistent march 3.5x slower, 0.257 vs 0.073 ms); constant-
    // address-space loads of a uniform address are scalar loads by construction.  The data is written before the
    // launch (host upload, prepass) and never during it.
    typedef const __attribute__((address_space(4))) double *TablePtr;
    typedef const __attribute__((address_space(4))) float *ConstF32Ptr;
    typedef const __attribute__((address_space(4))) int *ConstI32Ptr;
    const TablePtr tt = (TablePtr)(unsigned long long)a->t_table;
#ifndef GCFR_FRESH_LANE
#define GCFR_FRESH_LANE 1
#endif
    // (a kernel holds two instantiations of this function -- all-ones masks or not -- one after the other: computed from
    //  threadIdx the pixel's row / column are common subexpressions of both, hoisted in front of the first and kept alive
    //  across it for the second; a lane id the optimiser cannot see through is re-derived by each)
    const int lane = (GCFR_FRESH_LANE != 0) ? fresh_lane_id() : (int)(threadIdx.x & 63);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    // sample range of this wave
    constexpr bool KSPLIT = SPLIT == 1;
#ifndef GCFR_TRAIL
#define GCFR_TRAIL 1
#endif
#ifndef GCFR_HORIZON
#define GCFR_HORIZON 1
#endif
#ifndef GCFR_TRAIL_ALL_ONES
#define GCFR_TRAIL_ALL_ONES 1
#endif
    // the trailing loop, see the sample loop (all-ones masks: there is no mask work to save, but its termination test is the sharper one)
    constexpr bool TRAIL = (GCFR_TRAIL != 0) && !KSPLIT && !LDS && (!ALL_ONES || (GCFR_TRAIL_ALL_ONES != 0));
    constexpr bool ROUGH = MODE == kModeRough;  // the rough loop, see the sample loops
    const int chunk = KSPLIT ? (a->N + 3) >> 2 : a->N;
    const int k_lo = KSPLIT ? wave * chunk : 0;
    const int N = KSPLIT ? min(a->N, k_lo + chunk) : a->N;  // exclusive upper bound ("N" below)
#ifdef GCFR_COUNTERS
    unsigned cnt[kCntUsed] = {};
    unsigned cnt_since_body = 0, cnt_had_body = 0;  // visits since the last executed body / whether there was one
    // timeline record of this tile (tools/trace_timeline.py): constant 100 MHz clock + shader clock at entry
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime(), trace_c0 = __builtin_amdgcn_s_memtime();
#endif

    const int b = bl / L;
    const int l = bl - b * L;
    int r = qy * TILE_H + lane / TILE_W;
    int c = tx * TILE_W + (lane % TILE_W);
    const bool valid = (r < H) && (c < W);
    r = valid ? r : H - 1;
    c = valid ? c : W - 1;

    const size_t P = (size_t)H * W;
    const int Wp = W + 1;
    const size_t Pq = (size_t)(H + 1) * Wp;
    const __amdgpu_buffer_rsrc_t qr = make_rsrc(a->quad + (size_t)b * Pq, (int)(Pq * 16));
    const __amdgpu_buffer_rsrc_t mr =
        make_rsrc(a->mask + (size_t)(a->mask_batch == 1 ? 0 : b) * P, (int)P);

    // OWN: this lane's pixel is outside the mask (or outside the image): not marched.  Carried by `lane_last` = -1 from the
    // candidate range on (no register of its own): a lane without a candidate sample has only masked samples either way.
    const bool own_off = GCFR_M(17, false &&, ) OWN && (!valid || (buf_load_u8(mr, __mul24(r, W) + c) == 0));

    const ConstF32Ptr lp = (ConstF32Ptr)(unsigned long long)a->light_pt;
    const float Cx = lp[3 * bl + 0], Cy = lp[3 * bl + 1], Cz = lp[3 * bl + 2];
    const Box box = image_box(H, W);
    const LightCase lc = classify_light(Cx, Cy, box);
    const float halfWf = W / 2.0f, halfHf = H / 2.0f;
    const double halfW = W / 2.0, halfH = H / 2.0;
    const int halfWi = W / 2, halfHi = H / 2;

    const float x = (float)c - halfWf, y = halfHf - (float)r;
    const float zb = a->depth[(size_t)b * P + (size_t)r * W + c];
// census: end point (T8:378-465) + ray constants
    float Ex, Ey;
    end_point(x, y, Cx, Cy, box, lc, Ex, Ey);
    const float dxf = Ex - x, dyf = Ey - y;
    const float BCx = Cx - x, BCy = Cy - y, BCz = Cz - zb;
    const bool finite_ray = (dxf - dxf == 0.0f) && (dyf - dyf == 0.0f);
    const double x64 = x, y64 = y;
    const double dx64 = finite_ray ? (double)dxf : 0.0, dy64 = finite_ray ? (double)dyf : 0.0;
    // magic constants (see header comment); the y one is used as (My - sy)
    const double Mx = EVEN_HALF ? kRintMagic + halfW : kRintMagic;
    const double My = EVEN_HALF ? kRintMagic + halfH : kRintMagic;
    const int quad_origin = (Wp + 1) << 4;  // byte offset of texel (r=0, c=0)
#ifdef GCFR_AUDIT   // (see the counters: every claim about samples that are not evaluated, checked against their plain evaluation)
    RayConst arc;
    arc.H = H;
    arc.W = W;
    arc.halfW = halfW;
    arc.halfH = halfH;
    arc.x = x;
    arc.y = y;
    arc.zb = zb;
    arc.dx = dxf;
    arc.dy = dyf;
    arc.BCx = BCx;
    arc.BCy = BCy;
    arc.BCz = BCz;
    arc.x64 = x64;
    arc.y64 = y64;
    arc.dx64 = dx64;
    arc.dy64 = dy64;
    const __amdgpu_buffer_rsrc_t azr = make_rsrc(a->depth + (size_t)b * P, (int)(P * 4));
    bool audit_dead = false;  // pixels = mask: a lane that is not marched sees every sample as masked (by the option's definition)
    auto audit_S = [&](int k, bool &masked) -> float {  // sample k of this lane's ray, from the depth plane and the mask
        const float S = ray_sample(arc, (double)tt[k], azr, mr, masked);
        masked = masked || audit_dead;
        return S;
    };
    auto audit_count = [&](int slot, bool pred) { cnt[slot] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(pred)); };
    auto audit_masked = [&](int k_from, int k_to, bool claim) {  // "samples [k_from, k_to) of the lanes in `claim` are masked"
        for (int k = k_from; k < k_to; ++k) {
            bool m;
            (void)audit_S(k, m);
            audit_count(kCntAuditMaskedChecks, claim && finite_ray);
            audit_count(kCntAuditMaskedViol, claim && finite_ray && !m);
        }
    };
#endif
// census: give-up test (incl. nrm, c1)
    const ConstI32Ptr tfl = (ConstI32Ptr)(unsigned long long)a->tflag;  // the prepass' record about the sample table (scalar loads)
    const bool t_increasing = (a->N >= 2) && (tfl[kTfOk] != 0);  // checked by the prepass (see its table check)
    bool use_zb = !ROUGH && (a->zb != nullptr) && t_increasing;
    if (MODE == kModeFull && !use_zb)
        return true;  // (wave-uniform: facts about the launch) marched by the rough variant
    const int zrec = tfl[kTfStride];
    const int zls = use_zb ? (zrec & 0xff) : 3;
    const int zntw = (W >> zls) + 1;
    // (v_sqrt_f32 / v_rcp_f32, 1 ulp each: n and c1 only ever enter the BOUNDS -- the give-up heuristic, Kerr, the skip and the
    //  termination tests -- whose error terms K1 = ... + 1e-6 |c1| t and K2 r = (1e-6 n + ...) r budget sixteen ulp for each; the
    //  IEEE square root and division cost 27 instructions more per tile, round 5's census)
#ifdef GCFR_R04_FIXED_COST
    const float nrm = __builtin_sqrtf(BCx * BCx + BCy * BCy);
    const float c1 = BCz * ((dxf * BCx + dyf * BCy) / nrm);
#else
    const float nrm = __builtin_amdgcn_sqrtf(BCx * BCx + BCy * BCy);
    const float c1 = BCz * ((dxf * BCx + dyf * BCy) * __builtin_amdgcn_rcpf(nrm));
#endif
    // (round 4: the give-up test sits in FRONT of the candidate-range computation -- a tile that hands itself to the rough
    //  variant has then paid for the end point and this test only, not for the box / octagon clipping it would do twice)
    const float t_abs = __builtin_bit_cast(float, tfl[kTfTabs]);  // max(|tt[0]|, |tt[N-1]|)
    // Give-up test (a heuristic about WORK, never about results: without the bounds every group is marched).  A group
    // can only be skipped while the ray's height over the pixel, c1 t / n, exceeds what the surface band leaves open,
    // about half its width; on a surface rougher than the rays rise (an untrained network's depth) no test can ever
    // succeed and the tests, the bounds gathers and the termination checks are pure cost (round 1: -24 % against the
    // kernel without them at noise amplitude 400).  The band of the tile under the wave's own pixels stands for the
    // roughness of its neighbourhood: if for every lane the whole ray rises less than GCFR_GIVEUP_FACTOR band widths,
    // the wave marches this tile without the bounds machinery.
#ifndef GCFR_GIVEUP_FACTOR
#define GCFR_GIVEUP_FACTOR 0.5f
#endif
    if (use_zb) {
        const int own = __mul24((qy * TILE_H) >> zls, zntw) + ((tx * TILE_W) >> zls);
        const ConstF32Ptr rec = (ConstF32Ptr)(unsigned long long)a->zb + 4 * ((size_t)b * zb_slot(H, W) + own);
        const float band = rec[3] - rec[2];  // c_hi - c_lo (wave-uniform address: scalar loads)
        const bool hopeless = !(fabsf(c1) * t_abs >= GCFR_GIVEUP_FACTOR * nrm * band);  // (NaN / inf bands: hopeless)
        if (__builtin_amdgcn_ballot_w64(!hopeless) == 0ull) {
            if (MODE == kModeFull)
                return true;  // nothing has been marched or written: the caller runs the rough variant of this tile
            use_zb = false;
            GCFR_COUNT(kCntBoundsGivenUp, 1);
        }
    }

// census: candidate range: box + octagon clip, wave reductions
    float bestS = __builtin_inff();
    int besti = -1;
    float prevS = __builtin_inff();  // the running minimum replaced last (distance-tie resolution, see epilogue)
    int prevk = -1;
    bool any_masked = false;

    // Candidate sample range.  A sample can only be unmasked if its rounded cell lies inside the bounding
    // box of the mask's non-zero cells, i.e. if s(t) = start + t*delta lies inside that box inflated by
    // 0.5 (rint) plus a 0.01 safety margin.  That is an interval of t per lane; the union over the wave,
    // converted to sample indices with one step of slack either side, bounds the loop.  Everything outside
    // is masked for every lane, which only sets `any_masked` -- exact, and it removes the mask gathers of
    // rays that have left (or never reach) the face.  Requires the sample table to be monotone and
    // uniformly spaced to within half a step, which gcfr_sample_table guarantees.
    int k_begin = k_lo, k_end = N;  // [k_begin, k_end)
    // the pruning / skipping machinery below reasons about an increasing, uniform sample table inside [0, 1]
    // (gcfr_sample_table with dt > 0, the reference's np.arange); anything else marches every sample, which is
    // always right
    const int gz_lo_s = st.gz_lo_s, gz_nhi_s = st.gz_nhi_s;  // image depth range {z_min, -z_max} (sortable ints)
    int lane_last = (OWN && own_off) ? -1 : a->N - 1;  // last sample of this lane that can be unmasked (mask bounding box), see below
    if (t_increasing) {
        const int r_min = st.r_min, c_min = st.c_min, r_max = st.r_max, c_max = st.c_max;
        int lane_lo = a->N, lane_hi = -1;  // empty
        if (r_min != kBBoxInit && !(OWN && own_off)) {
            // (wave-uniform, and gfx950 has no scalar float unit: the integer part on the scalar unit, then one conversion and one fma
            //  each -- 0.5 (2 c - W) is exactly c - W/2)
            const float X0 = __builtin_fmaf(0.5f, (float)(2 * c_min - W), -GCFR_M(1, 0.49f, 0.51f)), X1 = __builtin_fmaf(0.5f, (float)(2 * c_max - W), GCFR_M(1, 0.49f, 0.51f));
            const float Y0 = __builtin_fmaf(0.5f, (float)(H - 2 * r_max), -GCFR_M(1, 0.49f, 0.51f)), Y1 = __builtin_fmaf(0.5f, (float)(H - 2 * r_min), GCFR_M(1, 0.49f, 0.51f));
            float ta = -3.0e38f, tb = 3.0e38f;
            bool empty = !finite_ray;
            // One slab: p + t dp inside [lo, hi].  dp == 0: no constraint on t if p is inside, no t at all if it is not.  Branch-free (round 5: as
            // `if (dp != 0) ... else ...` each of the four slabs was a divergent region of its own, ~19 VALU and ~18 scalar
            // instructions and branches apiece).  (v_rcp_f32, 1 ulp: the 0.01-pixel margin dwarfs it; four IEEE divisions cost ~50
            // VALU per wave)
            auto slab = [&](float p, float dp, float lo, float hi) {
                const bool still = dp == 0.0f;
                const float inv = __builtin_amdgcn_rcpf(dp);
                const float t1 = (lo - p) * inv, t2 = (hi - p) * inv;
                const float whole = ((p < lo) || (p > hi)) ? 3.0e38f : -3.0e38f;  // dp == 0: t in [-big, big], or the empty [big, -big]
                ta = fmaxf(ta, still ? whole : fminf(t1, t2));
                tb = fminf(tb, still ? -whole : fmaxf(t1, t2));
            };
            slab(x, dxf, X0, X1);
            slab(y, dyf, Y0, Y1);
            // ... and inside the mask's bounding OCTAGON (round 3): the cell's column + row and column - row lie within the
            // extents the prepass found, i.e. (s_x + W/2) +- (H/2 - s_y) does to within 1 (two roundings of 0.5) + 0.02.  For
            // an elliptical mask the octagon cuts four fifths of the box's corners: -18 % visited groups (tools/sim_octagon.py).
            if (st.mask_all_ones == 0) {  // (wave-uniform; an all-ones mask's octagon is its box)
                const int ws = W + H, wd = W - H;
                const float U0 = __builtin_fmaf(0.5f, (float)(2 * st.s_min - ws), -GCFR_M(2, 0.98f, 1.02f)), U1 = __builtin_fmaf(0.5f, (float)(2 * st.s_max - ws), GCFR_M(2, 0.98f, 1.02f));
                const float V0 = __builtin_fmaf(0.5f, (float)(2 * st.d_min - wd), -GCFR_M(2, 0.98f, 1.02f)), V1 = __builtin_fmaf(0.5f, (float)(2 * st.d_max - wd), GCFR_M(2, 0.98f, 1.02f));
                slab(x - y, dxf - dyf, U0, U1);
                slab(x + y, dxf + dyf, V0, V1);
            }
            if (!empty && ta <= tb) {
                const float t_first = __builtin_bit_cast(float, tfl[kTfTfirst]);  // (float)tt[0]
                const float inv_dt = __builtin_bit_cast(float, tfl[kTfInvDt]);   // (N - 1) / (tt[N-1] - tt[0]), v_rcp_f32
                const float ka = (ta - t_first) * inv_dt, kb = (tb - t_first) * inv_dt;
                // clamp in float first: ta / tb may be +-3e38
                lane_lo = (int)fminf(fmaxf(floorf(ka) - 1.0f, 0.0f), (float)a->N);
                lane_hi = (int)fmaxf(fminf(ceilf(kb) + 1.0f, (float)(a->N - 1)), -1.0f);
            }
        }
        // readfirstlane: the reductions are wave-uniform by construction, but only an SGPR tells the
        // compiler so -- with VGPR bounds the sample loop turns into a divergent loop (per-lane trip count,
        // vector loads of the sample table, +34 VGPRs: measured 20 % slower).
        lane_last = lane_hi;
#ifdef GCFR_AUDIT
        audit_dead = OWN && own_off;
        for (int k = 0; k < a->N; ++k) {  // the candidate range: every sample outside [lane_lo, lane_hi] is masked
            bool m;
            (void)audit_S(k, m);
            const bool claim = finite_ray && ((k < lane_lo) || (k > lane_hi));
            audit_count(kCntAuditMaskedChecks, claim);
            audit_count(kCntAuditMaskedViol, claim && !m);
        }
#endif
        // (round 5 tried both minima in one reduction -- signed 16-bit halves, v_pk_min_i16 behind each DPP move: more instructions, not
        //  fewer -- v_min_i32 takes the DPP operand itself, the packed minimum needs a move in front of it)
        const int w_lo = __builtin_amdgcn_readfirstlane(wave_min_i32(lane_lo));
        const int w_hi = -__builtin_amdgcn_readfirstlane(wave_min_i32(-lane_hi));
        const int nb = max(k_begin, w_lo), ne = min(k_end, w_hi + 1);
        any_masked = GCFR_M(24, false, (nb > k_begin) || (ne < k_end));  // some sample of this wave's range was pruned
        k_begin = nb;
        k_end = ne;
    }
// census: bounds set-up: Kerr, cap, safeS
    if (k_begin >= k_end)
        use_zb = false;  // no ray of this tile reaches the mask's box: nothing to march, so no bounds set-up either
    GCFR_COUNT(kCntTiles, 1);
    if (ROUGH)
        GCFR_COUNT(kCntBoundsGivenUp, 1);  // (every tile of the rough variant marches without the bounds, whatever the reason)
    GCFR_COUNT(kCntGroupsNominal, (N - k_lo + DEPTH - 1) / DEPTH);
    GCFR_COUNT(kCntSamplesInRange, k_end > k_begin ? k_end - k_begin : 0);

    // Depth-bound skip (exact).  For the sample point A = (s_k, z) of this ray,
    //     S_k >= Xx^2 + Xy^2 >= G^2,   G = n (z - zb) - BCz (BA_xy . u)/n,   u = BC_xy, n = |u|
    // (Cauchy-Schwarz on the two cross-product components that involve z): n (z - zb) is how far the sampled
    // surface is from the pixel's own depth and the second term how high the ray is there, both scaled by n.
    // (BA_xy . u)/n is t_k (d . u)/n to within 7e-4 (the 1e-4 offset and the f32 roundings of T8:480-487).  The
    // prepass' tile record bounds the surface by a band around a plane, z in a X + b Y + [c_lo, c_hi], valid at
    // the bilinear sample POSITION (see build_zbounds_block), so along one group of samples G is a linear
    // function of t inside [F_lo + t E, F_hi + t E] and |G| is at least `gap` below, evaluated at the group's
    // first and last sample.  The reference's bilinear weights are both 0 when a coordinate is integral, which
    // samples z = 0: that isolated value is tested too (gap0).  Kerr over-estimates every rounding between G
    // and the f32 S the body would compute (K1 + K2 r, r bounding |BA|'s components over the image, plus the
    // plane evaluation's terms); a lane votes "skip" only if the bound exceeds its running minimum by a further
    // 0.2 %, so a skipped sample could not have been taken and the minimum, its index and the tie predecessor
    // are what the full march gives.
    // (stride and fit of the bounds grid for groups of DEPTH samples: zb_log2_stride(), evaluated once by the prepass)
    const bool zfits = use_zb && ((zrec & 0x100) != 0);
    // With a checked table every sample lies on the segment pixel -> end point, i.e. inside the image, and the
    // stride was chosen so that a group's footprint fits the tile its lowest cell selects: no per-lane test.
    const bool zb_trusted = __builtin_amdgcn_readfirstlane((int)zfits) != 0;
    const __amdgpu_buffer_rsrc_t zr =
        make_rsrc(a->zb + (size_t)b * zb_slot(H, W), zb_slot(H, W) * (int)sizeof(float4));  // (records, then the horizon tables)
    // From here on the ray's f32 direction is RE-DERIVED from the f64 one where it is needed (the bounds test, the horizon
    // look-up, the tie re-march): (float)dx64 == dxf for every finite ray, and a non-finite ray takes no decision from either
    // (Kerr = +inf, c1 = NaN).  Two conversions per bounds test buy two registers -- the ones the six-wave kernel lacked with
    // the trailing loop in place, when the allocator spilled an f64 ray constant into the bodies instead.  (Laundered: or the
    // loop-invariant conversion is hoisted straight back into a register.)
    auto dir_f32 = [&](float &dxl, float &dyl) {
        if (WANT_ARGMIN) {  // (the five-wave training kernels have the two registers: round 5)
            dxl = finite_ray ? dxf : 0.0f;
            dyl = finite_ray ? dyf : 0.0f;
            return;
        }
        double dx_l = dx64, dy_l = dy64;
        asm volatile("" : "+v"(dx_l), "+v"(dy_l));
        dxl = (float)dx_l;
        dyl = (float)dy_l;
    };
    float Kerr = __builtin_inff();  // never skips
    // Early termination (exact).  Once the ray is above max(image depth maximum, 0) by more than the running
    // minimum allows (same bound as above, with the image-wide zmax instead of a tile's) and is still rising
    // (c1 > 0), no later sample of this lane can be taken.  The lane's `any_masked` no longer matters either if
    // its distance is certainly below the masked value 1e6 (safeS, wave-uniform).  A lane whose remaining
    // samples all lie outside the mask's bounding box is finished too (they are masked: any_masked).  When
    // every lane of the wave is finished the march stops -- it saves the mask gathers of the rest of the ray.
    float gz_cap = __builtin_inff();  // (wave-uniform) the cap: depth maximum over what a sample can read, >= 0; +inf: never finished by the bound
    float safeS = 0.0f;
    if (use_zb) {
        // r: bound on |BA|'s components over the whole image (x, y extent; depth range incl. the sampled 0)
        const float gz_lo = f32_unsortable(gz_lo_s), gz_hi = -f32_unsortable(gz_nhi_s);  // all-NaN image: +inf, -inf
        const float rr = fmaxf(fmaxf(fabsf(gz_lo - zb), fabsf(gz_hi - zb)), fmaxf(fabsf(zb), (float)max(H, W)));
        const float K1 = GCFR_M(3, 0.0f, 4e-3f * fabsf(BCz) + 1e-6f * fabsf(c1) * t_abs);
        const float K2 = GCFR_M(4, 0.0f, 1e-6f * nrm + 2e-7f * ((fabsf(BCx) + fabsf(BCy)) + fabsf(BCz)));
        // + the plane evaluation: position offsets (1e-4, t d vs the rounded BA_xy) times |a| + |b| <= 8, and the
        //   f32 roundings of a X + b Y (|.| <= 8 max(H, W)) at build and at test time
        const float K = __builtin_fmaf(K2, rr, K1) + GCFR_M(5, 0.0f, nrm * (1.2e-2f + 8e-6f * (float)max(H, W)));
        if ((nrm > 0.0f) && finite_ray && (K - K == 0.0f)) {
            Kerr = GCFR_M(23, 0.0f, K);
        }
        // the cap: the depth maximum over what an unmasked sample can read where the prepass built the horizon tables
        // (col_suf[0], see build_horizon_block), else over the whole image; either way >= 0 (NaN / inf: the test fails)
        if (a->hz_off >= 0) {  // col_suf's first entry: the maximum over every column (of each band)
            const ConstF32Ptr e = (ConstF32Ptr)(unsigned long long)a->zb + 4 * ((size_t)b * zb_slot(H, W) + zb_stride(H, W) + kHorizonDim);
            gz_cap = fmaxf(fmaxf(e[0], e[1]), fmaxf(e[2], e[3]));
        } else {
            gz_cap = GCFR_M(25, gz_hi, fmaxf(gz_hi, 0.0f));
        }
        gz_cap = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, gz_cap)));  // (an SGPR, for the compiler)
        // d = sqrt(S)/den < 1e6 for certain when S < 0.98e12 den^2; wave minimum -> SGPR
        const float den2 = (BCx * BCx + BCy * BCy) + BCz * BCz;
        const float s_lane = (den2 - den2 == 0.0f) ? GCFR_M(9, 1.02e12f, 0.98e12f) * den2 : 0.0f;
        safeS = f32_unsortable(__builtin_amdgcn_readfirstlane(wave_min_i32(f32_sortable(s_lane))));
    }

// census: loop: bounds record fetch
    // bounds of the cells a group can touch, given the rounded cells of its first and last sample:
    // floor(u) and ceil(u) lie in [rint(s) - 1, rint(s) + 1], so the extended indices are [min, max + 2].
    // A footprint the selected tile does not cover reads the sentinel record (-inf, +inf): it never skips.
    const int zb_sentinel = (zb_max_tiles(H, W) - 1) << 4;
    auto zb_fetch = [&](int ca, int ra, int cb, int rb) -> f32x4 {
        const int cmin = min(ca, cb), cmax = max(ca, cb), rmin = min(ra, rb), rmax = max(ra, rb);
        const int tj = cmin >> zls, ti = rmin >> zls;
        int off = (__mul24(ti, zntw) + tj) << 4;
        if (!zb_trusted) {  // (wave-uniform branch)
            const bool covered = (cmin >= 0) && (rmin >= 0) && (cmax <= W - 1) && (rmax <= H - 1) &&
                                 (cmax + 2 <= ((tj + 2) << zls) - 1) && (rmax + 2 <= ((ti + 2) << zls) - 1);
            // (no select on the loaded value: it would make the wave wait for the gather right here)
            off = covered ? off : zb_sentinel;
        }
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, off, 0, 0));
    };
    // the same record from the workgroup's LDS copy (bitmap first, records behind it)
    const int lds_zb_base = bitmap_stride_bytes(H, W);
    auto lds_load_b32 = [&](int byte_off) -> uint32_t {
        return *(const __attribute__((address_space(3))) uint32_t *)((__attribute__((address_space(3))) const char *)gcfr_lds_stage + byte_off);
    };
    auto lds_zb_fetch = [&](int ca, int ra, int cb, int rb) -> f32x4 {
        const int cmin = min(ca, cb), cmax = max(ca, cb), rmin = min(ra, rb), rmax = max(ra, rb);
        const int tj = cmin >> zls, ti = rmin >> zls;
        int off = (__mul24(ti, zntw) + tj) << 4;
        if (!zb_trusted) {  // (wave-uniform branch)
            const bool covered = (cmin >= 0) && (rmin >= 0) && (cmax <= W - 1) && (rmax <= H - 1) &&
                                 (cmax + 2 <= ((tj + 2) << zls) - 1) && (rmax + 2 <= ((ti + 2) << zls) - 1);
            off = covered ? off : zb_sentinel;
        }
        return *(const __attribute__((address_space(3))) f32x4 *)((__attribute__((address_space(3))) const char *)gcfr_lds_stage + lds_zb_base + off);
    };

// census: loop: mask prefetch (positions, rint, gathers)
    // Two-stage software pipeline.  Stage A (sample k+1): position, rounded cell, issue the mask byte
    // gather.  Stage B (sample k): if NO lane of the wave has an unmasked sample, the whole bilinear /
    // distance body is skipped -- masked samples only contribute "1e6" (T8:512), which `any_masked`
    // records.  The skip is wave-uniform (ballot -> scalar branch) and exact; on face-shaped masks more
    // than half of all wave-steps take it (rays that have left the face, background tiles).
    auto mask_offset = [&](double sx, double sy, int &col_r, int &row_r) -> int {  // T8:472-477, 510
        if (EVEN_HALF) {
            col_r = lo32(sx + Mx);  // rint(sx) + W/2
            row_r = lo32(My - sy);  // H/2 - rint(sy)
        } else {
            col_r = lo32(sx + Mx) + halfWi;
            row_r = halfHi - lo32(sy + My);
        }
        return __mul24(row_r, W) + col_r;
    };

    // Samples are processed in groups of DEPTH.  The group's mask bytes were gathered one group ahead;
    // if no lane has an unmasked sample anywhere in the group the whole group is skipped, otherwise the
    // DEPTH bodies run as straight-line code so their texel gathers are in flight together.  Indices
    // past N-1 are clamped to N-1: re-evaluating the last sample changes neither the minimum nor the
    // (first) argmin, so the tail needs no branch.
    auto clampk = [&](int k) { return k < k_end ? k : k_end - 1; };
#ifdef GCFR_AUDIT
    // "this lane's distance is certainly below the masked value 1e6" (where bestS < safeS lets a lane ignore its any_masked)
    auto audit_safe = [&](bool claim) {
        const float den_a = sqrt_rn_normal(((BCx * BCx + BCy * BCy) + BCz * BCz) + kEps4);
        const float d_a = sqrt_rn_normal(bestS) / den_a;
        audit_count(kCntAuditSafeViol, claim && finite_ray && !(d_a < kMaskedDistance));
    };
    // the wave stops in front of sample k_from: a lane finished by the bound (`by_bound`, its gd, the slack of the test) claims
    // S_k >= slack gd^2 for each of its later unmasked samples, a lane finished by its range (`by_last`) that they are all masked
    auto audit_finish = [&](int k_from, bool by_bound, float gd, float slack, bool by_last) {
        for (int k = k_from; k < k_end; ++k) {
            bool m;
            const float S = audit_S(k, m);
            audit_count(kCntAuditTermChecks, by_bound && !m);
            audit_count(kCntAuditTermViol, by_bound && !m && (S < gd * gd * slack));
            audit_count(kCntAuditMaskedChecks, !by_bound && by_last && finite_ray);
            audit_count(kCntAuditMaskedViol, !by_bound && by_last && finite_ray && !m);
        }
        audit_safe(by_bound);
    };
#endif
    struct Prefetched {  // what is gathered one group ahead: the group's mask bytes and its depth bounds
        uint32_t m[DEPTH];
        f32x4 z;  // {a, b, c_lo, c_hi}
    };
    // ALL_ONES: a mask without a single zero cell (the worst case of the mask-group skip: nothing is ever masked)
    // needs no mask gathers at all -- the prefetch then only locates the group's first and last cell for the bounds
    // record.  A compile-time variant of the whole tile function, chosen per tile by the caller: as a run-time
    // branch inside the prefetch it cost the common case 6 % (the sample loop's schedule falls apart around it).
    // The sample-table values of the group the next prefetch addresses are read one group EARLIER still, into SGPRs
    // (tq): an s_load issued right before its use stalls the wave for a scalar-cache round trip, twice per group as the
    // compiler scheduled it, and a wave that skips a group in ~250 cycles has nothing to hide that behind.
    double tq[DEPTH];
    double tc0 = 0.0, tc3 = 0.0;  // first / last table value of the group the next group() call consumes (SGPRs
                                  // are at 101 of ~106: carrying all four, so that the bodies need no s_load, spills)
    auto load_tq = [&](int kfirst) {
#pragma unroll
        for (int j = 0; j < DEPTH; ++j)
            tq[j] = tt[clampk(kfirst + j)];
    };
    auto pos_tq = [&](int j, double &sx, double &sy) {
        sx = x64 + tq[j] * dx64;  // T8:472 / 480 (f64, mul and add rounded separately)
        sy = y64 + tq[j] * dy64;
    };
    auto prefetch = [&](Prefetched &p) {  // (the group whose table values tq holds)
        int cj[DEPTH], rj[DEPTH];
        if (ALL_ONES) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j)
                p.m[j] = 1u;
            if (use_zb) {
                double px, py;
                pos_tq(0, px, py);
                (void)mask_offset(px, py, cj[0], rj[0]);
                pos_tq(DEPTH - 1, px, py);
                (void)mask_offset(px, py, cj[DEPTH - 1], rj[DEPTH - 1]);
                p.z = zb_fetch(cj[0], rj[0], cj[DEPTH - 1], rj[DEPTH - 1]);
            }
            return;
        }
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            double px, py;
            pos_tq(j, px, py);
            p.m[j] = buf_load_u8(mr, mask_offset(px, py, cj[j], rj[j]));
        }
        if (use_zb)
            p.z = zb_fetch(cj[0], rj[0], cj[DEPTH - 1], rj[DEPTH - 1]);
    };
// census: loop: termination test
    // One group: issue the next group's gathers into `nxt`, then consume `cur`.  Returns false when the wave is
    // finished (early termination).  The loop below alternates two buffers, so that the loaded registers are
    // consumed in place -- with a single buffer copied at the loop's back edge the compiler waits for the
    // gathers (s_waitcnt vmcnt(0)) at the END of the iteration that issued them, which exposes their whole
    // latency on every skipped group.
    // consume(): what happens to one group once its mask values `cm` (0 = masked) and its bounds record `cz` are known.
    // ta64 / tb64: the group's first / last table value; tn64: the next group's first one.
    // finish_check(): early termination, see Dcap.  false: no later sample of any lane of the wave can matter.
    auto finish_check = [&](int k0, double tn64, bool check_finished) -> bool {
        if (check_finished && use_zb && k0 + DEPTH < k_end) {
            const float tn = (float)tn64;  // tt[k0 + DEPTH]: the next group's first value
            // (n (cap - zb) + Kerr re-evaluated per test: as a per-lane constant it was one register more than the six-wave
            //  kernel has once the trailing loop is in place; Kerr = +inf where the bound is not valid: gd = -inf)
            float nrm_l = nrm;  // (laundered: or the loop-invariant sub-expression is hoisted back into a register and spilled)
            asm volatile("" : "+v"(nrm_l));
            // (Round 4 measured the trailing loop's per-ray cap -- two horizon-table gathers -- in THIS test too (VERDICT r03 item
            //  5a): bit-identical, and slower: -2.7 % on the bench faces, -3.0 % on the FFHQ fixtures, -3.5 % at B = 128
            //  (profiles/r04_pixels_mainhz_ab.txt).  The main loop rarely terminates, it hands over to the trailing loop.)
            const float gd = __builtin_fmaf(c1, tn, -(__builtin_fmaf(nrm_l, gz_cap, -(nrm_l * zb)) + Kerr));
            const float bS = bestS;
            const bool finished = (GCFR_M(11, true, (c1 > 0.0f)) && (gd > 0.0f) && (gd * gd * GCFR_M(7, GCFR_MUT_SLACK_VALUE, 0.998f) > bS) && (bS < safeS)) ||
                                  (lane_last < k0 + DEPTH GCFR_M(26, + 1, ));
            if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) {
                any_masked |= (lane_last < k0 + DEPTH GCFR_M(26, + 1, ));
                GCFR_COUNT(kCntEarlyExit, 1);
#ifdef GCFR_AUDIT
                audit_finish(k0 + DEPTH, GCFR_M(11, true, (c1 > 0.0f)) && (gd > 0.0f) && (gd * gd * GCFR_M(7, GCFR_MUT_SLACK_VALUE, 0.998f) > bS) && (bS < safeS),
                             gd, GCFR_M(7, GCFR_MUT_SLACK_VALUE, 0.998f), lane_last < k0 + DEPTH GCFR_M(26, + 1, ));
#endif
                return false;
            }
        }
        return true;
    };
// census: loop: depth-bound test
    // bound_cw(): the depth-bound test of one group for this lane -- true: no sample of the group can lower (or tie) the
    // lane's running minimum.  cz: the group's bounds record, ta64 / tb64: its first / last table value.
    auto bound_cw = [&](const f32x4 &cz, double ta64, double tb64, int k0) -> bool {
            GCFR_COUNT(kCntBoundTests, 1);
            const float ta = (float)ta64, tb = (float)tb64;  // tt[k0], tt[clampk(k0 + DEPTH - 1)]
            // (n zb re-multiplied per test from a laundered zb: hoisted out of the loops it is one more live register than the
            //  inference kernel has at six waves -- with the trailing loop in place the allocator spilled Dcap for it)
            float zb_l = zb;
            if (TRAIL && !WANT_ARGMIN)
                asm volatile("" : "+v"(zb_l));
            const float Qz = nrm * zb_l;
            const float Ta = c1 * ta, Tb = c1 * tb;
            const float Tlo = fminf(Ta, Tb), Thi = fmaxf(Ta, Tb);
            // surface band at the sample position s(t) = (x, y) + t d:  z in A0 + t A1 + [c_lo, c_hi], so
            // G(t) = n (z - zb) - c1 t  lies in  [F_lo + t E, F_hi + t E]: linear in t, extremes at the group's ends
            float dxl, dyl;
            dir_f32(dxl, dyl);
            const float A0 = __builtin_fmaf(cz.x, x, cz.y * y), A1 = __builtin_fmaf(cz.x, dxl, cz.y * dyl);
            const float E = __builtin_fmaf(nrm, A1, -c1);
            const float Flo = __builtin_fmaf(nrm, A0 + cz.z, -Qz), Fhi = __builtin_fmaf(nrm, A0 + cz.w, -Qz);
            const float eA = ta * E, eB = GCFR_M(29, ta, tb) * E;
            const float gap = fmaxf(Flo + fminf(eA, eB), -(Fhi + fmaxf(eA, eB)));  // > 0 iff the band stays clear of the ray
            const float gap0 = fmaxf(-Qz - Thi, Tlo + Qz);     // the same for the isolated value z = 0
            const float g = GCFR_M(10, gap, fminf(gap, gap0)) - Kerr;
#ifdef GCFR_AUDIT
            for (int j = 0; j < DEPTH; ++j) {  // g > 0: every unmasked sample of the group has S >= 0.998 g^2 (what a skip relies on)
                bool m;
                const float S = audit_S(clampk(k0 + j), m);
                const bool claim = (g > 0.0f) && !m;
                audit_count(kCntAuditBoundChecks, claim);
                audit_count(kCntAuditBoundViol, claim && (S < g * g * GCFR_M(6, 1.002f, 0.998f)));
                // how much of the budget the evaluation used: (the bound before Kerr - the true sqrt(S)) / Kerr, in 1/1000
                const float use = claim ? ((g + Kerr) - __builtin_sqrtf(S)) / Kerr : 0.0f;
                const int use_m = (int)fminf(fmaxf(use * 1000.0f, 0.0f), 1.0e6f);
                cnt[kCntAuditMaxUse] = max(cnt[kCntAuditMaxUse], (unsigned)(-wave_min_i32(-use_m)));
            }
#else
            (void)k0;
#endif
            return (g > 0.0f) && (g * g * GCFR_M(6, 1.002f, 0.998f) > bestS);
    };
// census: loop: sample body (bilinear, distance, minimum)
    // consume(): what happens to one group once its mask values `cm` (0 = masked) are known.  LAZY = false: `cz` is the
    // group's bounds record, tested here if some lane has an unmasked sample; LAZY = true (LDS-staged variant): the
    // caller has tested it already and passes the lane's verdict in `cw_in`.  ta64 / tb64: the group's first / last
    // table value; tn64: the next group's first one.
    // One sample's arithmetic behind its texel gather: bilinear depth (T8:480-494, f64), point A (T8:497-502), the squared
    // distance numerator (T8:503-509, f32, torch.cross's fma placement) and the running minimum with its tie predecessor.
    // Shared by the group bodies and the rough loop, so both evaluate the reference's rounding sequence with the same code.
    auto eval_sample = [&](int k, bool masked, double uxj, double uyj, double fxj, double fyj, const f32x4 &q) {
        const double gxd = __builtin_ceil(uxj), gyd = __builtin_ceil(uyj);
        const double wx0 = gxd - uxj, wx1 = uxj - fxj;
        const double wy0 = gyd - uyj, wy1 = uyj - fyj;
        const double zUL = q.x, zUR = q.y, zLL = q.z, zLR = q.w;
        const double up = zUL * wx0 + zUR * wx1;
        const double low = zLL * wx0 + zLR * wx1;
        const double zA = up * wy0 + low * wy1;
        const float Ax = (float)(uxj - halfW), Ay = (float)(halfH - uyj), Az = (float)zA;
        const float BAx = Ax - x, BAy = Ay - y, BAz = Az - zb;
        const float Xx = __builtin_fmaf(BAy, BCz, -(BAz * BCy));
        const float Xy = __builtin_fmaf(BAz, BCx, -(BAx * BCz));
        const float Xz = __builtin_fmaf(BAx, BCy, -(BAy * BCx));
        const float S = ((Xx * Xx + Xy * Xy) + Xz * Xz) + kEps4;
        const bool take = !masked && (S < bestS);
#ifdef GCFR_COUNTERS   // (round 4) per executed wave-sample: does ANY lane lower its minimum?  (the question an f32 pre-filter would ask)
        {
            const unsigned long long tk = __builtin_amdgcn_ballot_w64(take);
            cnt[kCntWaveSamples] += 1u;
            cnt[kCntWaveSamplesTaken] += (tk != 0ull) ? 1u : 0u;
            cnt[kCntLaneTakes] += (unsigned)__builtin_popcountll(tk);
        }
#endif
        if (WANT_ARGMIN) {
            prevS = take ? bestS : prevS;
            prevk = take ? besti : prevk;
            besti = take ? k : besti;
        }
        bestS = take ? S : bestS;
    };
// census: loop: group bookkeeping (ballots, branches)
    bool all_lazy = false;  // (wave-uniform) set by consume(): the group just consumed could have been skipped without its mask
    auto consume = [&](int k0, const uint32_t (&cm)[DEPTH], const f32x4 &cz, double ta64, double tb64, double tn64,
                       bool check_finished, auto lazy, bool cw_in) -> bool {
        constexpr bool LAZY = decltype(lazy)::value;
        if (TRAIL)
            all_lazy = false;
        if (!LAZY)
            GCFR_COUNT(kCntGroupsVisited, 1);
#ifdef GCFR_COUNTERS
        ++cnt_since_body;
#endif
        const bool dead = OWN && (lane_last < 0);  // (OWN: a lane whose own pixel is outside the mask sees every sample as masked)
        bool none = true;
#pragma unroll
        for (int j = 0; j < DEPTH; ++j) {
            none = none && (cm[j] == 0);
            any_masked |= (cm[j] == 0);
        }
        none = none || dead;
        bool run_body = __builtin_amdgcn_ballot_w64(!none) != 0ull;
        if (LAZY) {
            run_body = __builtin_amdgcn_ballot_w64(!none && !cw_in) != 0ull;
        } else if (run_body && use_zb) {
            const bool cannot_win = bound_cw(cz, ta64, tb64, k0);
            run_body = __builtin_amdgcn_ballot_w64(!none && !cannot_win) != 0ull;
            if (TRAIL)  // (see the trailing loop below) nothing of this group mattered to any lane, masked or not
                all_lazy = !run_body && __builtin_amdgcn_ballot_w64(!((cannot_win && GCFR_M(18, true, (bestS < safeS))) || (lane_last < k0))) == 0ull;
        }
        // samples of the group evaluated together (texel gathers in flight): one at a time in the six-wave inference
        // variant (fewer live registers -> forced occupancy, see the __global__ wrappers), two in the argmin variant, the
        // whole group in the k-split variant, whose launches are tiny and latency-bound
        // (the argmin variant runs at five waves per SIMD and has the registers for two gathers in flight: +5 % on smooth and on
        //  rough depth, round 3; the six-wave inference variant at five waves with two in flight: -2 ... -3 %)
#ifndef GCFR_INFER_BODY_CHUNK
#define GCFR_INFER_BODY_CHUNK 1   // (-DGCFR_MARCH_WAVES_PER_EU=5 -DGCFR_INFER_BODY_CHUNK=2, re-measured in round 5 on the bench faces: one batch at a
                                  //  time +1.8 % (1.100 against 1.080 T), four in flight -0.8 % (2.180 against 2.197 T); four waves with two or
                                  //  four in flight lose both ways: profiles/r05_body_chunk_ab.txt)
#endif
        constexpr int GCFR_BODY_CHUNK = KSPLIT ? DEPTH : (WANT_ARGMIN && DEPTH >= 2 ? 2 : GCFR_INFER_BODY_CHUNK);  // (KSPLIT here: SPLIT == 1 only)
        if (run_body) {
          GCFR_COUNT(kCntBodies, 1);
#ifdef GCFR_COUNTERS
          if (!cnt_had_body)
              cnt[kCntVisitsBeforeFirstBody] += cnt_since_body - 1;
          cnt_had_body = 1;
          cnt_since_body = 0;
#endif
#ifdef GCFR_COUNTERS
#pragma unroll
          for (int j = 0; j < DEPTH; ++j)
              cnt[kCntLaneSamples] += (unsigned)__builtin_popcountll(__builtin_amdgcn_ballot_w64(cm[j] != 0));
#endif
#pragma unroll
// census: loop: sample body (bilinear, distance, minimum)
          for (int h0 = 0; h0 < DEPTH; h0 += GCFR_BODY_CHUNK) {
            // phase 1: positions and texel gathers for the whole group (all in flight together)
            double ux[DEPTH], uy[DEPTH], fxd[DEPTH], fyd[DEPTH];
            f32x4 qv[DEPTH];
#pragma unroll
            for (int j = h0; j < h0 + GCFR_BODY_CHUNK && j < DEPTH; ++j) {
                // the group's first and last table value are already in SGPRs (ta64 / tb64): no s_load, no wait (+1.2 %
                // for the fused kernels; the march-only kernels, one register short at six waves, spill 16 B with it
                // and lose 3 %: they keep the loads)
                constexpr bool CARRIED = FUSE_SHADE || WANT_ARGMIN;
                const double tj = (CARRIED && j == 0) ? ta64 : ((CARRIED && j == DEPTH - 1) ? tb64 : (double)tt[clampk(k0 + j)]);
                const double sx = x64 + tj * dx64, sy = y64 + tj * dy64;  // T8:472 / 480 (mul and add rounded separately)
                ux[j] = (sx + halfW) - 0.0001;  // unrounded position (T8:480-487)
                uy[j] = (halfH - sy) - 0.0001;
                fxd[j] = __builtin_floor(ux[j]);
                fyd[j] = __builtin_floor(uy[j]);
                const int fx = (int)fxd[j], fy = (int)fyd[j];  // may be -1: the quad grid has that row / column
                const int texel = __mul24(fy, Wp) + fx;
                // NB: bit-cast the whole vector.  Indexing the builtin's result element-wise makes this
                // hipcc narrow the load to ONE dword (all four corners alias) -- caught in the ISA.
                qv[j] = __builtin_bit_cast(
                    f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, (texel << 4) + quad_origin, 0, 0));
            }
            // phase 2: bilinear depth, point A, squared distance numerator, running minimum
#pragma unroll
            for (int j = h0; j < h0 + GCFR_BODY_CHUNK && j < DEPTH; ++j)
                eval_sample(clampk(k0 + j), (cm[j] == 0) || dead, ux[j], uy[j], fxd[j], fyd[j], qv[j]);
          }
        }
// census: loop: group bookkeeping (ballots, branches)
        return finish_check(k0, tn64, check_finished);
    };
    auto group = [&](int k0, const Prefetched &cur, Prefetched &nxt, bool check_finished) -> bool {
        const double ta64 = tc0, tb64 = tc3;  // first / last table value of THIS group (tq of the previous call)
        tc0 = tq[0];
        tc3 = tq[DEPTH - 1];
        prefetch(nxt);                 // group k0 + DEPTH
        load_tq(k0 + 2 * DEPTH);       // ... and the table values of the one after it
        return consume(k0, cur.m, cur.z, ta64, tb64, tc0, check_finished, std::false_type{}, false);
    };

// census: loop: LDS-staged variant
    // LDS-staged variant (round 3).  The workgroup's image -- its mask as a BITMAP and its depth-bounds
    // records -- was copied into LDS by the four waves at kernel entry (stage_lds, global_load_lds_dwordx4); what the
    // global variant gathers one group ahead through the texture path (four 1-byte mask gathers, each occupying the
    // addressers like a full-width load, and one 16-byte record per lane) is here four ds_read_b32 + one ds_read_b128 of
    // the group ITSELF: LDS answers in ~100 cycles, so there is no second register buffer and no two-groups-ahead table
    // bookkeeping, and the texture path carries only the bodies' texel gathers.  Same cells, same records, same
    // arithmetic after them: bit-identical.
    auto group_lds = [&](int k0, bool check_finished) -> bool {
        const double ta64 = tq[0], tb64 = tq[DEPTH - 1];  // (tq holds THIS group's table values)
        uint32_t cm[DEPTH];
        int cj[DEPTH], rj[DEPTH], cell[DEPTH];
        GCFR_COUNT(kCntGroupsVisited, 1);
        // (1) the bounds test FIRST: it needs the first and the last sample's cells only.  The mask decides nothing for a
        //     lane that cannot win AND already holds a minimum that is certainly below the masked value 1e6 (bestS <
        //     safeS: its `any_masked` no longer matters, see the early termination) -- if that is every lane of the wave,
        //     the group is over without a single mask lookup: the two middle positions, the four bitmap reads and the
        //     mask bookkeeping are never computed.  Only the LDS variant can order it this way: the global variant has to
        //     start its mask gathers a group ahead, before it knows anything.
        bool cw = false;
        if (use_zb) {
            double px, py;
            pos_tq(0, px, py);
            cell[0] = mask_offset(px, py, cj[0], rj[0]);
            pos_tq(DEPTH - 1, px, py);
            cell[DEPTH - 1] = mask_offset(px, py, cj[DEPTH - 1], rj[DEPTH - 1]);
            const f32x4 cz = lds_zb_fetch(cj[0], rj[0], cj[DEPTH - 1], rj[DEPTH - 1]);
            cw = bound_cw(cz, ta64, tb64, k0);
            if (__builtin_amdgcn_ballot_w64(!(cw && (bestS < safeS))) == 0ull) {
#ifdef GCFR_AUDIT
                audit_safe(cw);
#endif
                load_tq(k0 + DEPTH);
                return finish_check(k0, tq[0], check_finished);
            }
        }
        // (2) some lane may still win (or has no minimum yet): the group's mask bits, then the bodies
        if (ALL_ONES) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j)
                cm[j] = 1u;
        } else {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                if (!(use_zb && (j == 0 || j == DEPTH - 1))) {
                    double px, py;
                    pos_tq(j, px, py);
                    cell[j] = mask_offset(px, py, cj[j], rj[j]);  // row * W + col; W % 32 == 0: 32 cells of a row per dword
                }
                cm[j] = lds_load_b32((cell[j] >> 3) & ~3);
            }
        }
        load_tq(k0 + DEPTH);           // the next group's table values (scalar loads, answered while this group runs)
        if (!ALL_ONES) {
#pragma unroll
            for (int j = 0; j < DEPTH; ++j)
                cm[j] = __builtin_amdgcn_ubfe(cm[j], (uint32_t)cj[j], 1u);  // bit (col mod 32): the offset operand uses 5 bits
        }
        return consume(k0, cm, f32x4{}, ta64, tb64, tq[0], check_finished, std::true_type{}, cw);
    };

// census: loop: LDS-staged variant
    if (LDS) {
        __syncthreads();  // the staged image is complete (every wave of the workgroup gets here exactly once)
        if (k_begin < k_end)
            load_tq(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += 2 * DEPTH) {
            if (!group_lds(k0, false))
                break;
            if (k0 + DEPTH >= k_end)
                break;
            if (!group_lds(k0 + DEPTH, true))
                break;
        }
    } else if (ROUGH) {
// census: loop: rough loop
        // Rough loop (round 4).  A tile that marches WITHOUT the depth bounds -- it gave them up (a surface rougher than its
        // rays rise: an untrained network's depth, 7,259 of 8,192 tiles at noise amplitude 400), the caller switched them off, or
        // the sample table is not one the bounds reason about -- executes every sample of its candidate range: nothing is
        // decided a group ahead, so nothing is gathered a group ahead.  Per sample ONE position (the main loop computes it
        // twice: for the mask prefetch and again in the body), the mask byte and the texel gathered together, CH samples
        // in flight, no group bookkeeping, no termination test -- the candidate range's end [k_begin, k_end) IS the wave's
        // last sample that can be unmasked.  Same arithmetic per sample (eval_sample): bit-identical.
        constexpr int CH = WANT_ARGMIN ? GCFR_ROUGH_CHUNK_ARGMIN : GCFR_ROUGH_CHUNK;
        double tr[CH];
        auto load_tr = [&](int kfirst) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                tr[j] = tt[clampk(kfirst + j)];
        };
        if (k_begin < k_end)
            load_tr(k_begin);
        for (int k0 = k_begin; k0 < k_end; k0 += CH) {
            GCFR_COUNT(kCntRoughSamples, min(CH, k_end - k0));
            double ux[CH], uy[CH], fxd[CH], fyd[CH];
            f32x4 qv[CH];
            uint32_t mk[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const double tj = tr[j];
                const double sx = x64 + tj * dx64, sy = y64 + tj * dy64;  // T8:472 / 480 (mul and add rounded separately)
                int cr, rr;
                const int moff = mask_offset(sx, sy, cr, rr);
                mk[j] = ALL_ONES ? 1u : buf_load_u8(mr, moff);
                ux[j] = (sx + halfW) - 0.0001;  // unrounded position (T8:480-487)
                uy[j] = (halfH - sy) - 0.0001;
                fxd[j] = __builtin_floor(ux[j]);
                fyd[j] = __builtin_floor(uy[j]);
                const int texel = __mul24((int)fyd[j], Wp) + (int)fxd[j];  // (-1: the quad grid has that row / column)
                qv[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(qr, (texel << 4) + quad_origin, 0, 0));
            }
            load_tr(k0 + CH);  // the next iteration's table values (scalar loads, answered while this one runs)
            // (Round 4 measured skipping the evaluation when NO lane of the wave has an unmasked sample among these CH -- holes in
            //  the mask, the gap between octagon and face: a ballot and a branch per iteration cost more than they save, -2.4 % at
            //  noise 400, -2 % on the FFHQ masks with noise; profiles/r04_rough_ab.txt.)
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const bool masked = (mk[j] == 0) || (OWN && (lane_last < 0));
                any_masked |= masked;
                eval_sample(clampk(k0 + j), masked, ux[j], uy[j], fxd[j], fyd[j], qv[j]);
            }
        }
    } else {
// census: loop: main loop driver
    Prefetched bufA, bufB;
    bufA.z = bufB.z = f32x4{0.0f, 0.0f, -__builtin_inff(), __builtin_inff()};
    const int k_first = k_begin;
    if (k_first < k_end) {
        load_tq(k_first);
        tc0 = tq[0];
        tc3 = tq[DEPTH - 1];
        prefetch(bufA);
        load_tq(k_first + DEPTH);
    }
    int k_trail = k_end;  // where the trailing loop takes over (k_end: nowhere)
    f32x4 zcur = bufA.z;  // ... and the bounds record of that group
    for (int k0 = k_first; k0 < k_end; k0 += 2 * DEPTH) {
        if (!group(k0, bufA, bufB, false))
            break;
        if (k0 + DEPTH >= k_end)
            break;
        if (TRAIL && all_lazy) {
            k_trail = k0 + DEPTH;
            zcur = bufB.z;
            break;
        }
        if (!group(k0 + DEPTH, bufB, bufA, true))  // the termination test runs every other group (it costs ~18 VALU)
            break;
        if (TRAIL && all_lazy && k0 + 2 * DEPTH < k_end) {
            k_trail = k0 + 2 * DEPTH;
            zcur = bufA.z;
            break;
        }
    }
// census: loop: trailing loop
    // Trailing loop (round 3).  72 % of the groups a tile visits on face-shaped data come AFTER its last body: the rays run on
    // above the surface, the bounds test rejects group after group, and each of them still paid for four f64 sample
    // positions and four mask gathers one group ahead.  Once the test of a group has come out "nothing here matters to
    // any lane" -- every lane either cannot win and holds a minimum certainly below the masked value 1e6 (bestS < safeS:
    // its `any_masked` is irrelevant, see the early termination), or has left the mask's bounding box for good (its
    // samples are masked: any_masked) -- the wave predicts the same for the groups that follow (measured on the bench
    // faces before the horizon tables shortened this loop: 4,655 of 8,192 tiles make the prediction, 69,849 groups -- 55 %
    // of all visits -- are walked this way, 3,069 break it) and walks them with the bounds records alone: first and last cell of the group, one record
    // gather a group ahead, the test; no mask, no middle positions.  A group whose test does NOT come out that way gets
    // its mask bytes on the spot (one exposed gather latency) and the ordinary treatment.  Exact: a group is only ever
    // skipped on the strength of the same test the main loop applies, and then the mask decides nothing for it.
    if (TRAIL) {
        if (k_trail < k_end)
            GCFR_COUNT(kCntTrailEnter, 1);
        auto record_of = [&](double t_first, double t_last) -> f32x4 {  // the bounds record of the group with these table values
            int ca, ra, cb, rb;
            (void)mask_offset(x64 + t_first * dx64, y64 + t_first * dy64, ca, ra);
            (void)mask_offset(x64 + t_last * dx64, y64 + t_last * dy64, cb, rb);
            return zb_fetch(ca, ra, cb, rb);
        };
        // here: zcur = the record of group k_trail, tc0 / tc3 = its first / last table value, tq = the table values of the group after it
        // Termination in this loop: the same test as finish_check(), with the cap taken from the horizon tables where the
        // prepass built them -- the maximum over the columns AND over the rows the rest of this lane's ray can still touch
        // (running maxima from the next sample's cell towards the side the ray is heading for; the slack that covers the bilinear
        // corners and the f32 position is built into the tables: build_horizon_block), instead of the image-wide maximum.
        for (int k0 = k_trail; k0 < k_end; k0 += DEPTH) {
            const double ta64 = tc0, tb64 = tc3;
            tc0 = tq[0];
            tc3 = tq[DEPTH - 1];
            const f32x4 znext = record_of(tc0, tc3);  // group k0 + DEPTH, one group ahead
            load_tq(k0 + 2 * DEPTH);
            const bool check = (((k0 - k_trail) / DEPTH) & 1) != 0;  // every other group, as in the main loop (every group: measured level, -0.5 %)
            // (the tables' offset is read from the kernel arguments at every use -- a scalar load -- rather than kept in a register)
            const int hz_off = (GCFR_HORIZON != 0) && check && (k0 + DEPTH < k_end) ? launder(a)->hz_off : -1;
            const bool check_hz = hz_off >= 0;
            const float tn = (float)tc0;  // the next group's first table value
            const bool cw = bound_cw(zcur, ta64, tb64, k0);
            const bool gone = lane_last < k0 GCFR_M(27, + 1, );
            GCFR_COUNT(kCntGroupsVisited, 1);
            if (__builtin_amdgcn_ballot_w64(!((cw && GCFR_M(18, true, (bestS < safeS))) || gone)) == 0ull) {
                any_masked |= gone;
                GCFR_COUNT(kCntTrailSkips, 1);
#ifdef GCFR_AUDIT
                audit_safe(cw && !gone);                              // (the lane's any_masked is not updated: it must not matter)
                audit_masked(k0, min(k0 + DEPTH, k_end), gone);      // (a lane that is gone: its samples from here on are masked)
#endif
#ifdef GCFR_COUNTERS
                ++cnt_since_body;
#endif
                if (check_hz) {
                    // (the two gathers are issued here, not a group ahead: eight more live registers across the bounds test
                    //  spill in the six-wave kernel; the wave waits for them once per two groups of this loop)
                    float dxl, dyl;
                    dir_f32(dxl, dyl);
                    const int ci = (int)__builtin_floorf(__builtin_fmaf(tn, dxl, x)), ri = (int)__builtin_floorf(-__builtin_fmaf(tn, dyl, y));
                    constexpr int C = kHorizonDim / 2, M = kHorizonDim - 1;
                    const int ic = (dxl >= 0.0f) ? kHorizonDim + min(max(ci + C, 0), M) : min(max(ci + C, 0), M);               // col_suf : col_pre
                    const int ir = (dyl > 0.0f) ? 2 * kHorizonDim + min(max(ri + C, 0), M) : 3 * kHorizonDim + min(max(ri + C, 0), M);  // row_pre : row_suf
                    const f32x4 zc4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, hz_off + (ic << 4), 0, 0));
                    const f32x4 zr4 = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zr, hz_off + (ir << 4), 0, 0));
                    const float zc = fmaxf(fmaxf(zc4.x, zc4.y), fmaxf(zc4.z, zc4.w)), zr_ = fmaxf(fmaxf(zr4.x, zr4.y), fmaxf(zr4.z, zr4.w));
                    float nrm_l = nrm;  // (laundered, see finish_check)
                    asm volatile("" : "+v"(nrm_l));
                    const float Dc = __builtin_fmaf(nrm_l, fminf(zc, zr_), -(nrm_l * zb)) + Kerr;  // (Kerr = inf: never finished)
                    const float gd = __builtin_fmaf(c1, tn, -Dc);
                    const bool past = lane_last < k0 + DEPTH GCFR_M(26, + 1, );
                    const bool finished = (GCFR_M(11, true, (c1 > 0.0f)) && (gd > 0.0f) && (gd * gd * GCFR_M(8, GCFR_MUT_SLACK_VALUE, 0.998f) > bestS) && (bestS < safeS)) || past;
                    if (__builtin_amdgcn_ballot_w64(!finished) == 0ull) {
                        any_masked |= past;
                        GCFR_COUNT(kCntEarlyExit, 1);
#ifdef GCFR_AUDIT
                        audit_finish(k0 + DEPTH, GCFR_M(11, true, (c1 > 0.0f)) && (gd > 0.0f) && (gd * gd * GCFR_M(8, GCFR_MUT_SLACK_VALUE, 0.998f) > bestS) && (bestS < safeS),
                                     gd, GCFR_M(8, GCFR_MUT_SLACK_VALUE, 0.998f), past);
#endif
                        break;
                    }
                } else if (!finish_check(k0, tc0, check)) {
                    break;
                }
            } else {  // some lane needs this group: its mask bytes now, then as in the main loop
                GCFR_COUNT(kCntTrailLeave, 1);
                uint32_t cm[DEPTH];
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    const double tj = (j == 0) ? ta64 : ((j == DEPTH - 1) ? tb64 : (double)tt[clampk(k0 + j)]);
                    int cj, rj;
                    cm[j] = ALL_ONES ? 1u : buf_load_u8(mr, mask_offset(x64 + tj * dx64, y64 + tj * dy64, cj, rj));
                }
                if (!consume(k0, cm, zcur, ta64, tb64, tc0, check, std::true_type{}, cw))
                    break;
                zcur = record_of(tc0, tc3);  // (fetched again rather than kept in registers across the bodies)
                continue;
            }
            zcur = znext;
        }
    }
    }

// census: k-split combine
    if (KSPLIT) {  // combine the four waves' partial results for this tile
        __shared__ float sS[4][64], sPS[4][64];
        __shared__ int sK[4][64], sPK[4][64];
        __shared__ uint8_t sM[4][64];
        sS[wave][lane] = bestS;
        sK[wave][lane] = besti;
        sPS[wave][lane] = prevS;
        sPK[wave][lane] = prevk;
        sM[wave][lane] = any_masked ? 1 : 0;
        __syncthreads();
        if (wave != 0)
            return false;
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            const float Sq = sS[q][lane];
            const bool take = Sq < bestS;  // strict: the earlier quarter keeps ties (first minimum, T8:514)
            // predecessor of a new best from quarter q: q's own predecessor if it already beat the running
            // best (it is part of the global chain of running minima), otherwise the running best it replaces
            const float PSq = sPS[q][lane];
            const bool local_pred = PSq < bestS;
            prevS = take ? (local_pred ? PSq : bestS) : prevS;
            prevk = take ? (local_pred ? sPK[q][lane] : besti) : prevk;
            bestS = take ? Sq : bestS;
            besti = take ? sK[q][lane] : besti;
            any_masked |= (sM[q][lane] != 0);
        }
    }

// census: epilogue: distance finish, tie, masked value, bonus
    // (BCx laundered: otherwise the prologue's BCx^2 + BCy^2 is kept in a register across the whole sample loop for this
    //  one use -- the last value the six-wave build spilled)
    float BCx_e = BCx;
    asm volatile("" : "+v"(BCx_e));
    // (sqrt_rn_normal: the correctly rounded square root for arguments >= 1e-4 -- both carry the reference's + 0.0001 -- see gcfr_device.hpp)
    const float den = sqrt_rn_normal(((BCx_e * BCx_e + BCy * BCy) + BCz * BCz) + kEps4);
    float d = sqrt_rn_normal(bestS) / den;
    // torch.min (T8:514) returns the FIRST index of the minimal distance: see first_tied_sample.
    if (WANT_ARGMIN) {
        const bool tie = GCFR_M(16, false &&, ) (prevk >= 0) && (sqrt_rn_normal(prevS) / den == d);
        if (__builtin_amdgcn_ballot_w64(tie) != 0ull) {  // rare; wave-uniform branch
            GCFR_COUNT(kCntTieRemarch, 1);
            RayConst rc;
            rc.H = H;
            rc.W = W;
            rc.halfW = halfW;
            rc.halfH = halfH;
            rc.x = x;
            rc.y = y;
            rc.zb = zb;
            dir_f32(rc.dx, rc.dy);
            rc.BCx = BCx;
            rc.BCy = BCy;
            rc.BCz = BCz;
            rc.x64 = x64;
            rc.y64 = y64;
            rc.dx64 = dx64;
            rc.dy64 = dy64;
            const int first = first_tied_sample(rc, a->t_table, make_rsrc(a->depth + (size_t)b * P, (int)(P * 4)), mr,
                                                tie, prevk, den, d);
            besti = tie ? first : besti;
        }
    }
    if (any_masked && !(d < kMaskedDistance)) {
        d = kMaskedDistance;
        besti = -1;
    }
    if (OWN && (lane_last < 0)) {  // not marched (or a ray that never reaches the mask: the same value either way): the masked value
        d = kMaskedDistance;
        besti = -1;
    }
    if (!finite_ray)  // (after the line above: a non-finite ray has no candidate range either, and is NaN with or without the option)
        d = __builtin_nanf("");
    const EpiPtr ep = launder((EpiPtr)&a->epi);
    const bool inside = (Cx >= ep->bx_lo) && (Cx <= ep->bx_hi) && (Cy >= ep->by_lo) && (Cy <= ep->by_hi);
    if (inside)
        d = d + ep->bonus;
// census: epilogue: pixel re-derivation, min_dist / argmin stores
    // The pixel's row / column / validity are RE-DERIVED here from a fresh lane id instead of being kept live across the
    // sample loop: at the forced six waves per SIMD (80 VGPRs) they were exactly what the register allocator spilled
    // (r, c and the 64-bit pixel index: 16-20 B of scratch per lane, stored before the loop and reloaded after it --
    // cheap in time, but the scratch arena of every resident wave is written back to HBM once per launch: +15 MB).
    {
        const int lane_e = fresh_lane_id();
        const int r_e = qy * TILE_H + lane_e / TILE_W, c_e = tx * TILE_W + (lane_e % TILE_W);
        const bool valid_e = (r_e < H) && (c_e < W);
        r = valid_e ? r_e : H - 1;
        c = valid_e ? c_e : W - 1;
        if (!valid_e)
            return false;
    }
    {
        const size_t pix = (size_t)r * W + c;
        const size_t o = (size_t)bl * P + pix;
        ep->min_dist[o] = d;
        if (WANT_ARGMIN)
            ep->argmin[o] = besti;
// census: epilogue: normals load / stencil call, normals_out store
        if (FUSE_SHADE) {
            float n[3];
            const float *normals = ep->normals;
            if (normals) {
                const float *nrm = normals + (size_t)b * 3 * P + pix;
                n[0] = nrm[0];
                n[1] = nrm[P];
                n[2] = nrm[2 * P];
            } else {  // normals fused: 3x3 depth stencil, same device function as normals_fwd_kernel
                NormalsArgs na = {};  // (field by field: the source lives in the constant address space)
                na.H = H;
                na.W = W;
                na.inv_fx = ep->nrm.inv_fx;
                na.inv_fy = ep->nrm.inv_fy;
                na.cx = ep->nrm.cx;
                na.cy = ep->nrm.cy;
                na.z_offset = ep->nrm.z_offset;
                na.negate_y = ep->nrm.negate_y;
                unit_normal(na, a->depth + (size_t)b * P, r, c, n);
                float *normals_out = ep->normals_out;
                if (normals_out && l == 0) {
                    float *no = normals_out + (size_t)b * 3 * P + pix;
                    no[0] = n[0];
                    no[P] = n[1];
                    no[2 * P] = n[2];
                }
            }
// census: epilogue: shading call, composite, stores (T8:517-522)
            const Shaded sh = shade_pixel(x, y, zb, n[0], n[1], n[2], Cx, Cy, Cz, ep->ambient[bl], ep->intensity, d);
            float *shadow_w = ep->shadow_w, *full = ep->full, *final_shading = ep->final_shading;
            if (shadow_w)
                shadow_w[o] = sh.w;
            if (full)
                full[o] = sh.full;
            if (final_shading)
                final_shading[o] = sh.fin;
            const float *alb = ep->albedo + (size_t)b * 3 * P + pix;
            float *ren = ep->rendered + (size_t)bl * 3 * P + pix;
#pragma unroll
            for (int ch = 0; ch < 3; ++ch)  // T8:519-522
                ren[ch * P] = alb[ch * P] * sh.fin;
        }
    }
// census: counters (counting build only)
#ifdef GCFR_COUNTERS
    if (a->counters && lane == 0 && !(KSPLIT && wave != 0)) {
        cnt[kCntVisitsAfterLastBody] += cnt_since_body;  // (tiles without any body: all their visits)
#ifndef GCFR_TRACE_ONLY   // (ten same-address atomics per tile cost ~20 ns each: they distort the timeline)
#pragma unroll
        for (int i = 0; i < kCntUsed; ++i) {
            if (i == kCntAuditMaxUse)
                atomicMax(a->counters + i, (unsigned long long)cnt[i]);
            else
                atomicAdd(a->counters + i, (unsigned long long)cnt[i]);
        }
#endif
        // per-tile record after the GCFR_N_COUNTERS tallies: {t0, t1 (100 MHz), shader cycles, hw ids | work}
        unsigned long long *rec = a->counters + GCFR_N_COUNTERS + 4 * ((size_t)(bl * a->tiles_y + qy) * a->tiles_x + tx);
        const unsigned hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID: wave, simd, cu, sh, se
        const unsigned xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
        rec[0] = trace_t0;
        rec[1] = __builtin_amdgcn_s_memrealtime();
        rec[2] = __builtin_amdgcn_s_memtime() - trace_c0;
        rec[3] = (unsigned long long)hw_id | ((unsigned long long)(xcc_id & 0xf) << 32) |
                 ((unsigned long long)(cnt[kCntBodies] & 0xfff) << 36) | ((unsigned long long)(cnt[kCntGroupsVisited] & 0xfff) << 48);
    }
#endif
    return false;
}

// census: kernel entry: grid mapping, instantiation choice (march_grid)
// The __global__ entry points of the march.  Occupancy is forced (the register allocator would settle at
// 95-99 VGPRs = 5 waves/SIMD): with the group body evaluated one sample at a time (GCFR_BODY_CHUNK = 1) the
// inference variant fits six waves per SIMD, which beats the 119-VGPR / 4-wave build that kept four gathers in
// flight per body by 8 % at B=8 on four streams and by 11 % at B=64.  The argmin variant carries three more loop
// registers and is best at five waves (at six: 28 B of scratch for -3 %).  Neither spills at its occupancy
// (tests/test_kernel_resources.py).  More waves do not pay even when they nearly fit -- end of round 2: 7 waves / 72
// VGPRs / 12 B of scratch -12 %, 8 waves / 40 B -30 % -- because the texture addressers and the L1 are the second
// resource near their limit (Roofline).  (tools/build_variant.sh + tools/ab.sh, tools/exp_grazing.py)
#ifndef GCFR_MARCH_WAVES_PER_EU
#define GCFR_MARCH_WAVES_PER_EU 6
#endif
#ifndef GCFR_MARCH_ARGMIN_WAVES_PER_EU
#define GCFR_MARCH_ARGMIN_WAVES_PER_EU 5
#endif

// Grid schedule: one workgroup = four horizontally adjacent tiles (one per wave), 3-D grid x = tile-quad column,
// y = tile row, z = (image, light) -- no integer divisions, dispatch order image-major with row-major tiles.
template <int TILE_W, bool EVEN_HALF, bool WANT_ARGMIN, int DEPTH, bool FUSE_SHADE, bool LDS = false, bool OWN = false>
__device__ __forceinline__ void march_grid(ArgPtr a)
{
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int tx = (int)blockIdx.x * 4 + wave;
    const int bl = a->bl_offset + (int)blockIdx.z;
    const bool want_z = (a->zb != nullptr);
    const ImageStats st = reduce_image_stats(a, bl / a->L, threadIdx.x & 63, want_z);
    if (LDS)
        stage_lds(a, bl / a->L, st.mask_all_ones == 0);  // every wave of the workgroup copies its share, tile or no tile
    if (tx >= a->tiles_x) {
        if (LDS)
            __syncthreads();  // (the barrier the marching waves pass in front of their sample loop)
        return;  // (without LDS staging the waves of a workgroup never synchronise)
    }
    // (the all-ones test is wave-uniform -- a fact about the mask, valid for any sample table; no pixel is outside such a mask)
    constexpr int FULL = (LDS || GCFR_ROUGH == 0) ? kModeInline : kModeFull;
    bool rough;
    if (st.mask_all_ones != 0)
        rough = march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, 0, true, LDS, false, FULL>(a, bl, (int)blockIdx.y, tx, st);
    else
        rough = march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, 0, false, LDS, OWN, FULL>(a, bl, (int)blockIdx.y, tx, st);
    if (FULL == kModeFull && GCFR_M(15, false &&, ) __builtin_amdgcn_readfirstlane((int)rough) != 0) {  // the tile marches without the depth bounds
        // (the image statistics are reduced AGAIN, through a laundered pointer -- a dozen scalar loads: kept alive across the
        //  bounds variant for this call they were eleven more SGPRs than the kernel has, spilled into a VGPR's lanes for the
        //  whole kernel: one register less in the main loops and a v_readlane at every use, -1.2 % on the bench faces)
        //  (the five-wave training kernels have the SGPRs to spare in VGPR lanes, and re-plan worse with the second reduction:
        //   20-44 B of scratch; they pass the statistics on)
        const ArgPtr a2 = WANT_ARGMIN ? a : launder(a);
        const int bl2 = WANT_ARGMIN ? bl : a2->bl_offset + (int)blockIdx.z;
        const ImageStats st2 = WANT_ARGMIN ? st : reduce_image_stats(a2, bl2 / a2->L, threadIdx.x & 63, false);
        if (st2.mask_all_ones != 0)
            march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, 0, true, false, false, kModeRough>(a2, bl2, (int)blockIdx.y, tx, st2);
        else
            march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, 0, false, false, OWN, kModeRough>(a2, bl2, (int)blockIdx.y, tx, st2);
    }
}

// (SCHED is the schedule the kernel was built for; the product has the grid only -- the parameter keeps the kernel
//  names of rounds 1-2 in profiles and tools: shadow_fwd_quad_kernel<16, true, 4, true, 0>.)
enum { kSchedGrid = 0 };

template <int SCHED, int TILE_W, bool EVEN_HALF, bool WANT_ARGMIN, int DEPTH, bool FUSE_SHADE, bool LDS = false, bool OWN = false>
__device__ __forceinline__ void march_dispatch()
{
    static_assert(SCHED == kSchedGrid, "the product library has the grid schedule only");
    march_grid<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, LDS, OWN>(kernel_args());
}

template <int TILE_W, bool EVEN_HALF, int DEPTH, bool FUSE_SHADE, int SCHED>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_MARCH_WAVES_PER_EU, GCFR_MARCH_WAVES_PER_EU))) void shadow_fwd_quad_kernel(ShadowQuadArgs)
{
    march_dispatch<SCHED, TILE_W, EVEN_HALF, false, DEPTH, FUSE_SHADE>();
}
template <int TILE_W, bool EVEN_HALF, int DEPTH, bool FUSE_SHADE, int SCHED>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_MARCH_ARGMIN_WAVES_PER_EU, GCFR_MARCH_ARGMIN_WAVES_PER_EU))) void shadow_fwd_quad_argmin_kernel(ShadowQuadArgs)
{
    march_dispatch<SCHED, TILE_W, EVEN_HALF, true, DEPTH, FUSE_SHADE>();
}
// pixels = mask (gcfr_options.pixels = 1): the training march that leaves out the pixels outside the mask (OWN, see march_tile)
template <int TILE_W, bool EVEN_HALF, int DEPTH, bool FUSE_SHADE, int SCHED>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_MARCH_ARGMIN_WAVES_PER_EU, GCFR_MARCH_ARGMIN_WAVES_PER_EU))) void shadow_fwd_quad_argmin_own_kernel(ShadowQuadArgs)
{
    march_dispatch<SCHED, TILE_W, EVEN_HALF, true, DEPTH, FUSE_SHADE, false, true>();
}
// LDS-staged variants (see group_lds in march_tile): the same kernels with the image's mask bitmap and bounds records in LDS
template <int TILE_W, bool EVEN_HALF, int DEPTH, bool FUSE_SHADE, int SCHED>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_MARCH_WAVES_PER_EU, GCFR_MARCH_WAVES_PER_EU))) void shadow_fwd_quad_lds_kernel(ShadowQuadArgs)
{
    march_dispatch<SCHED, TILE_W, EVEN_HALF, false, DEPTH, FUSE_SHADE, true>();
}
template <int TILE_W, bool EVEN_HALF, int DEPTH, bool FUSE_SHADE, int SCHED>
__global__ __launch_bounds__(256)
__attribute__((amdgpu_waves_per_eu(GCFR_MARCH_ARGMIN_WAVES_PER_EU, GCFR_MARCH_ARGMIN_WAVES_PER_EU))) void shadow_fwd_quad_argmin_lds_kernel(ShadowQuadArgs)
{
    march_dispatch<SCHED, TILE_W, EVEN_HALF, true, DEPTH, FUSE_SHADE, true>();
}
// k-split (tiny launches, one or two images): latency-bound, four gathers in flight per body, occupancy as it falls;
// grid x = tile column, y = tile row, z = (image, light)
template <int TILE_W, bool EVEN_HALF, bool WANT_ARGMIN, int DEPTH, bool FUSE_SHADE>
__global__ __launch_bounds__(256) void shadow_fwd_quad_ksplit_kernel(ShadowQuadArgs)
{
    const ArgPtr a = kernel_args();
    const int bl = a->bl_offset + (int)blockIdx.z;
    const ImageStats st = reduce_image_stats(a, bl / a->L, threadIdx.x & 63, a->zb != nullptr);
    march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, 1>(a, bl, (int)blockIdx.y, (int)blockIdx.x, st);
}


// ----------------------------------------------------------------------------------------------
// launchers: pick the kernel for (parity of W/2 and H/2, argmin wanted, schedule) and enqueue it
// ----------------------------------------------------------------------------------------------
enum Schedule { kGrid = kSchedGrid, kKSplit, kGridLds, kGridOwn };

template <int TILE_W, int DEPTH, bool FUSE>
static void launch_quad4(const ShadowQuadArgs &a, bool even_half, bool want_argmin, Schedule sch, dim3 grid,
                         hipStream_t st, unsigned lds_bytes)
{
#define GCFR_LAUNCH(KERNEL, ...) hipLaunchKernelGGL((KERNEL<TILE_W, __VA_ARGS__>), grid, dim3(256), 0, st, a)
#define GCFR_LAUNCH_LDS(KERNEL, ...) hipLaunchKernelGGL((KERNEL<TILE_W, __VA_ARGS__>), grid, dim3(256), lds_bytes, st, a)
#define GCFR_LAUNCH_SCHED(SCHED)                                                    \
    do {                                                                            \
        if (even_half) {                                                            \
            if (want_argmin)                                                        \
                GCFR_LAUNCH(shadow_fwd_quad_argmin_kernel, true, DEPTH, FUSE, SCHED);  \
            else                                                                    \
                GCFR_LAUNCH(shadow_fwd_quad_kernel, true, DEPTH, FUSE, SCHED);      \
        } else {                                                                    \
            if (want_argmin)                                                        \
                GCFR_LAUNCH(shadow_fwd_quad_argmin_kernel, false, DEPTH, FUSE, SCHED); \
            else                                                                    \
                GCFR_LAUNCH(shadow_fwd_quad_kernel, false, DEPTH, FUSE, SCHED);     \
        }                                                                           \
    } while (0)
    if (sch == kKSplit) {
        if (even_half) {
            if (want_argmin)
                GCFR_LAUNCH(shadow_fwd_quad_ksplit_kernel, true, true, DEPTH, FUSE);
            else
                GCFR_LAUNCH(shadow_fwd_quad_ksplit_kernel, true, false, DEPTH, FUSE);
        } else {
            if (want_argmin)
                GCFR_LAUNCH(shadow_fwd_quad_ksplit_kernel, false, true, DEPTH, FUSE);
            else
                GCFR_LAUNCH(shadow_fwd_quad_ksplit_kernel, false, false, DEPTH, FUSE);
        }
    } else if (sch == kGridLds) {
        if constexpr (TILE_W == 16 && DEPTH == 4) {  // (the default shape: the one the LDS-staged kernels are built for)
            if (even_half) {
                if (want_argmin)
                    GCFR_LAUNCH_LDS(shadow_fwd_quad_argmin_lds_kernel, true, DEPTH, FUSE, kSchedGrid);
                else
                    GCFR_LAUNCH_LDS(shadow_fwd_quad_lds_kernel, true, DEPTH, FUSE, kSchedGrid);
            } else {
                if (want_argmin)
                    GCFR_LAUNCH_LDS(shadow_fwd_quad_argmin_lds_kernel, false, DEPTH, FUSE, kSchedGrid);
                else
                    GCFR_LAUNCH_LDS(shadow_fwd_quad_lds_kernel, false, DEPTH, FUSE, kSchedGrid);
            }
        }
    } else if (sch == kGridOwn) {  // (argmin wanted: checked by the caller)
        if (even_half)
            GCFR_LAUNCH(shadow_fwd_quad_argmin_own_kernel, true, DEPTH, FUSE, kSchedGrid);
        else
            GCFR_LAUNCH(shadow_fwd_quad_argmin_own_kernel, false, DEPTH, FUSE, kSchedGrid);
    } else {
        GCFR_LAUNCH_SCHED(kSchedGrid);
    }
#undef GCFR_LAUNCH_SCHED
#undef GCFR_LAUNCH_LDS
#undef GCFR_LAUNCH
}

template <int TILE_W, int DEPTH>
static void launch_quad3(const ShadowQuadArgs &a, bool even_half, bool want_argmin, Schedule sch, dim3 grid,
                         hipStream_t st, unsigned lds_bytes)
{
    if (a.epi.rendered)
        launch_quad4<TILE_W, DEPTH, true>(a, even_half, want_argmin, sch, grid, st, lds_bytes);
    else
        launch_quad4<TILE_W, DEPTH, false>(a, even_half, want_argmin, sch, grid, st, lds_bytes);
}

// The translation units.  launch_march_<tile width>_<group>() is launch_quad3<tile width, group>; each is defined by one
// compilation of gcfr_march_unit.hip (build.py: MARCH_UNITS) and holds the kernels of that shape.
typedef void (*MarchUnitFn)(const ShadowQuadArgs &a, bool even_half, bool want_argmin, Schedule sch, dim3 grid,
                            hipStream_t st, unsigned lds_bytes);
#define GCFR_MARCH_UNIT_NAME_(TW, G) launch_march_##TW##_##G
#define GCFR_MARCH_UNIT_NAME(TW, G) GCFR_MARCH_UNIT_NAME_(TW, G)
#define GCFR_DECLARE_MARCH_UNIT(TW, G)                                                                              \
    void GCFR_MARCH_UNIT_NAME(TW, G)(const ShadowQuadArgs &a, bool even_half, bool want_argmin, Schedule sch, dim3 grid, \
                                     hipStream_t st, unsigned lds_bytes);
GCFR_DECLARE_MARCH_UNIT(16, 4)
#ifndef GCFR_FAST_BUILD   // (development builds have the default shape only: tools/build_variant.sh ... -DGCFR_FAST_BUILD)
GCFR_DECLARE_MARCH_UNIT(16, 2)
GCFR_DECLARE_MARCH_UNIT(16, 1)
GCFR_DECLARE_MARCH_UNIT(8, 4)
GCFR_DECLARE_MARCH_UNIT(8, 2)
GCFR_DECLARE_MARCH_UNIT(8, 1)
GCFR_DECLARE_MARCH_UNIT(32, 4)
GCFR_DECLARE_MARCH_UNIT(32, 2)
GCFR_DECLARE_MARCH_UNIT(32, 1)
GCFR_DECLARE_MARCH_UNIT(64, 4)
GCFR_DECLARE_MARCH_UNIT(64, 2)
GCFR_DECLARE_MARCH_UNIT(64, 1)
#endif

}  // namespace gcfr
