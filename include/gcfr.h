/*
 * gcfr.h -- C ABI of libgcfr_hip.so: the MI355X (gfx950) implementation of GeomConsistentFR's
 * render block (ray-marched soft shadow + Lambertian shading + compositing).
 *
 * The reference (andrewhou1/GeomConsistentFR) has NO operator/plugin/FFI interface for this path:
 * the block is ~170 inline lines of torch ops at the end of RelightNet.forward
 * (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:352-524, "T8"; inference variants
 * test_relight_single_image.py:326-505 "S1", test_relight_single_image_lighting_transfer.py:325-514
 * "SLT").  The entry points below are therefore the seams a maintainer would cut at T8:352; each one
 * cites the reference lines it replaces.  INTEGRATION.md shows the ctypes binding and the edit to
 * RelightNet.forward.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller (hipMalloc / torch tensor storage),
 *     contiguous, planes in NCHW order; nothing is allocated, freed or synchronised inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work;
 *   - re-entrant and thread-safe: the library keeps NO process-wide mutable state -- every knob and hook is
 *     an argument (`gcfr_options`), so two host threads may launch on two streams with different options;
 *     return value: GCFR_OK or a negative gcfr_status;
 *   - shapes: B images, L lights per image, H rows, W columns, N samples per ray.
 *     Pixel (r,c) has image-plane coordinates x = c - W/2, y = H/2 - r           (T8:51-55).
 */
#ifndef GCFR_H
#define GCFR_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum gcfr_status {
    GCFR_OK = 0,
    GCFR_ERR_INVALID_ARGUMENT = -1, /* null pointer / non-positive or unsupported dimension */
    GCFR_ERR_LAUNCH = -2            /* hipGetLastError() != hipSuccess after the launch       */
} gcfr_status;

/* Library / build identification: "gcfr-hip <version> gfx950". */
const char *gcfr_version(void);

/*
 * ABI revision of THIS header.  It changes whenever an entry point's argument list or a struct layout changes
 * (every symbol keeps its name, so a stale library would otherwise be called with shifted arguments).  A binding
 * compares gcfr_abi_version() of the library it loaded with the GCFR_ABI_VERSION it was written against and refuses
 * to proceed on a mismatch (geomconsistentfr_amd/_lib.py does).
 *   3: round 3 -- gcfr_inference_images_u8 gained `mask_f32`; gcfr_abi_version itself; the metrics entry points.
 *   4: round 4 -- gcfr_options gained `pixels`.
 *   5: round 5 -- gcfr_options gained `phase` (the prepass as its own enqueue); gcfr_copy_probe.
 *   6: round 6 -- gcfr_options lost the dead `schedule` / `tile_order` fields (sizeof 64 -> 56: `phase` had gone into what
 *      was tail padding, so revision 5's struct_size could not tell a revision-4 caller from a revision-5 one -- this one
 *      can); gcfr_inference_images_u8 gained `L` (relit images per photograph: many lights per face).
 * struct_size guards the struct's SIZE only: a field added into padding does not change it.  The revision check is
 * therefore mandatory for every binding, and every entry point validates struct_size before it reads any other field.
 */
#define GCFR_ABI_VERSION 6
int32_t gcfr_abi_version(void);

/*
 * Per-call options of the forward entry points (HOST struct, read during the call only; NULL = defaults).
 * With ONE exception (`pixels`, off by default) nothing here changes a result bit: the knobs select among kernels /
 * schedules that are bit-identical (tests/test_gpu_parity.py asserts it for every combination), the hooks only observe.
 */
#define GCFR_N_COUNTERS 28
typedef struct gcfr_options {
    uint32_t struct_size;      /* sizeof(gcfr_options) of the caller's build; a mismatch is GCFR_ERR_INVALID_ARGUMENT */
    int32_t tile_w;            /* pixels per tile row: 8, 16, 32 or 64 (a wave marches a tile_w x 64/tile_w tile); 0 = auto */
    int32_t group;             /* samples per skip group: 1, 2 or 4; 0 = auto (4) */
    int32_t ksplit;            /* split each tile's sample range over the 4 waves of a workgroup: 0, 1, -1 = auto by launch size */
    int32_t depth_bound_skip;  /* exact depth-bound group skip: 0, 1, -1 = auto (on) */
    int32_t lds_stage;         /* the march's workgroups copy their image's mask (as a bitmap) and depth-bounds records into
                                  LDS and read them there instead of gathering them through the texture path: 0 off, 1 on
                                  (wherever the shape allows: W % 32 == 0, 26 KiB per workgroup, default tile and group),
                                  -1 = auto */
    void *event_start;         /* hipEvent_t recorded on `stream` immediately before the march kernel, or NULL */
    void *event_stop;          /* hipEvent_t recorded immediately after it, or NULL */
    uint64_t *counters;        /* DEVICE array of GCFR_N_COUNTERS + 4 * (number of tiles) u64: the march kernel adds its work
                                  counts (executed groups, bound tests, ...; tools/count_work.py) to the first
                                  GCFR_N_COUNTERS and writes a 4-word timeline record per tile behind them
                                  (tools/trace_timeline.py); only a library built with -DGCFR_COUNTERS touches it
                                  (gcfr_version() then ends in "+counters"), else ignored.  With -DGCFR_AUDIT as well ("+audit")
                                  the last eight tallies count the claims the march makes about samples it does not evaluate and
                                  the ones a plain evaluation of those samples contradicts (tools/audit.py) */
    int32_t pixels;            /* WHICH PIXELS are marched.  0 (default; also -1): every pixel, as the reference does
                                  (T8:371-515 computes minimum_distance for all H x W pixels).
                                  1 ("mask"; DEVIATES from the reference's returned tensors, opt-in): pixels whose OWN mask cell
                                  is zero are not marched -- they get the masked value the reference assigns to a ray without
                                  an unmasked sample (minimum_distance 1e6, T8:512; argmin -1), hence shadow_mask_weights 1,
                                  final_shading = full_shading, rendered_images = albedo x full_shading there -- and a tile
                                  without an unmasked pixel does no march at all.  Every other pixel is bit-identical to
                                  pixels = 0.  Why it is safe for the training script: each consumer of the block's outputs
                                  multiplies them by that same mask (T8:619, 633, 641, 643: `rendered_images * masks` in the
                                  reconstruction, adversarial and DSSIM terms), so the losses are bit-equal and the masked-out
                                  pixels receive a zero upstream gradient either way; `shadow_mask_weights` / `full_shading` /
                                  `ambient_light` are returned by T8:524 but used by no loss.  Not for callers that
                                  look at the shadow OUTSIDE the mask (the inference scripts write the un-masked shadow image
                                  only after multiplying by the mask as well, S8:603-608).  Honoured by the workspace path when
                                  `argmin` is requested (the training forward; halves its march on face-shaped masks);
                                  GCFR_ERR_INVALID_ARGUMENT without a workspace or without `argmin`. */
    int32_t phase;             /* WHICH HALF of a workspace call is enqueued (gcfr_shadow_fwd, gcfr_render_fwd,
                                  gcfr_render_from_depth_fwd).  0 (default; also -1): both -- the prepass launch (depth repack, mask
                                  statistics, depth bounds, horizon tables, light preparation, sample-table check: everything that
                                  depends on depth, mask, light_raw and t_table only) and the march launch behind it on `stream`.
                                  1: the prepass only.  2: the march only, on a workspace that a phase-1 call WITH THE SAME ARGUMENTS
                                  AND OPTIONS filled (the caller orders the two: same stream, or an event between two streams).
                                  Bit-identical to phase 0 by construction: the same two launches with the same arguments, enqueued by
                                  two calls.  Why: in RelightNet.forward the depth head, the light and the mask exist before the albedo
                                  decoder has run (T8:226-350), so the prepass can run on a side stream under the decoder's
                                  convolutions and the march then starts without a dependent launch in front of it.  A phase-1 call
                                  reads only: depth, mask_u8, light_raw (where the entry point has it), t_table; writes: workspace,
                                  unit_out / light_pt_out (where the entry point has them).  The remaining pointers are not touched
                                  and may be NULL (the albedo does not exist yet).  GCFR_ERR_INVALID_ARGUMENT without a workspace. */
} gcfr_options;

/* Fills `opt` with the defaults (struct_size set, every knob "auto", no hooks). */
void gcfr_options_default(gcfr_options *opt);

/*
 * Sample fractions t_k along the pixel->light segment, HOST side helper.
 * Replaces np.arange(t0, ., dt) at T8:468 (S1:445, SLT:451) with numpy's value rule
 * t_k = t0 + k*((t0+dt)-t0), all f64.  Writes n doubles to `out_host`; upload it and pass the
 * device copy as `t_table`.
 */
int gcfr_sample_table(double t0, double dt, int32_t n, double *out_host);

/*
 * Light preparation.  Replaces T8:357-363 (clamp_z = 1, clamp_min = 0; SLT:332 uses 0.16) and
 * S1:332-336 (clamp_z = 0).
 *   light_raw     (n,3) f32   SL_lin2[...,1:4] or target_lighting
 *   unit_out      (n,3) f32   unit_light_direction
 *   light_pt_out  (n,3) f32   incident_light_points = light_distance * unit (T8:362)
 */
int gcfr_light_prep(const float *light_raw, int32_t n, int32_t clamp_z, float clamp_min,
                    float light_distance, float *unit_out, float *light_pt_out, void *stream);

/*
 * Ray march: minimum point-to-line distance over the sample table.  Replaces T8:371-515
 * (S1:349-496, SLT:355-504): slopes/intercepts, the nine-way end-point branch, clamp, the
 * (N,2,H,W) f64 sample grids, five gathers, bilinear depth, cross product distance, mask, min.
 *   depth       (B,H,W) f32      c2_o_depth, already x100 (T8:350)
 *   mask_u8     (MB,H,W) u8      1 where the reference's mask != 0; MB = mask_batch = B (T8:510)
 *                                or 1 (one mask shared by all images, S1:488)
 *   light_pt    (B,L,3) f32      from gcfr_light_prep
 *   t_table     (N) f64          from gcfr_sample_table (device copy).  Any table gives the reference's
 *                                results; the workspace path's pruning / skipping only engages on a table its
 *                                prepass finds increasing, inside [0, 1] and uniform to 0.1 %
 *   bonus       added to the minimum when the light's (x,y) lies inside bonus_box
 *               = {x_lo, x_hi, y_lo, y_hi} (S1:495-496: box = image, bonus = 5; SLT:503-504);
 *               training form: bonus = 0 (bonus_box may be NULL).  bonus_box is a HOST pointer.
 *   min_dist    (B,L,H,W) f32 out   minimum_distance (T8:515)
 *   argmin      (B,L,H,W) i32 out   index of the minimising sample (saved for backward); may be NULL
 *   opt         per-call options (host pointer) or NULL for the defaults; see gcfr_options
 *   workspace   device scratch of >= gcfr_shadow_workspace_bytes(B,H,W) bytes, 16-byte aligned, or NULL.
 *               With a workspace the depth maps are first repacked into 2x2-neighbourhood texels
 *               (one 16-byte gather per ray-step instead of four 4-byte gathers) and a coarse grid of
 *               depth bounds lets the march skip sample groups that provably cannot lower a
 *               pixel's running minimum; since round 3 the prepass also leaves, per image, running column / row
 *               maxima of the depth an unmasked sample can read ("horizon tables", shapes with W % 4 == 0 and
 *               H, W <= 1024), against which a ray that has passed its last candidate stops marching; results are
 *               bit-identical to the NULL-workspace path, only faster.  Contents are scratch: rewritten by every
 *               call, nothing in it needs initialising.
 * Supported: 2 <= H,W <= 4096, even; 1 <= N <= 4096.
 */
size_t gcfr_shadow_workspace_bytes(int32_t B, int32_t H, int32_t W);

int gcfr_shadow_fwd(const float *depth, const uint8_t *mask_u8, int32_t mask_batch,
                    const float *light_pt, int32_t B, int32_t L, int32_t H, int32_t W, int32_t N,
                    const double *t_table, float bonus, const float *bonus_box, float *min_dist,
                    int32_t *argmin, void *workspace, size_t workspace_bytes, void *stream,
                    const gcfr_options *opt);

/*
 * Soft-shadow transfer + Lambert shading + composite.  Replaces T8:364-369 and T8:517-522.
 *   normals     (B,3,H,W) f32    depth_to_normals(depth+offset, K) with y negated (T8:353-354);
 *                                 re-normalised inside (T8:365)
 *   depth       (B,H,W) f32;  albedo (B,3,H,W) f32;  light_pt (B,L,3) f32;  ambient (B,L) f32
 *   min_dist    (B,L,H,W) f32    from gcfr_shadow_fwd
 *   intensity   directional_intensity (T8:46: 0.5; SLT:20: 0.41)
 * outputs (any may be NULL except rendered):
 *   shadow_w    (B,L,H,W) f32    shadow_mask_weights = 1 - 4e^-d/(1+e^-d)^2        (T8:517)
 *   full        (B,L,H,W) f32    full_shading = ambient + I*max(n.l, 0)            (T8:366-369)
 *   final       (B,L,H,W) f32    final_shading = w*full + (1-w)*ambient            (T8:518)
 *   rendered    (B,L,3,H,W) f32  rendered_images = albedo * final                  (T8:519-522)
 */
int gcfr_shade_fwd(const float *normals, const float *depth, const float *albedo,
                   const float *light_pt, const float *ambient, const float *min_dist, int32_t B,
                   int32_t L, int32_t H, int32_t W, float intensity, float *shadow_w, float *full,
                   float *final_shading, float *rendered, void *stream);

/*
 * Surface normals from depth.  Replaces the kornia call + sign flip at T8:353-354 (SLT:325: offset 1410,
 * focal 700).  kornia 0.4.1 is un-vendored: this restates its published algorithm (unproject with the
 * camera matrix, normalised 3x3 Sobel with replicate padding, cross, normalise) -- parity UNPINNED.
 *   depth (B,H,W) f32;  fx, fy, cx, cy from intrinsic_matrix (T8:571-577);  z_offset added in f32
 *   negate_y: 1 = apply T8:354;  normals (B,3,H,W) f32 out, unit length
 * Numerical contract (also of the normals the fused epilogue of gcfr_render_from_depth_fwd computes and of `normals_out`):
 * NOT bit-reproducible against another evaluation order -- the reference's chain is f64 with an unspecified conv2d
 * summation order, and the kernels use reciprocals / v_rsq + Newton steps instead of IEEE divisions -- but within 4 f32
 * ulp of the f64 restatement (oracle/normals_restatement.py) after rounding to f32; a non-finite depth cell makes
 * exactly the normals non-finite whose clamped 3 x 3 neighbourhood contains it (tests/test_gpu_normals.py).
 */
int gcfr_normals_fwd(const float *depth, int32_t B, int32_t H, int32_t W, double fx, double fy, double cx,
                     double cy, float z_offset, int32_t negate_y, float *normals, void *stream);
/* Backward of gcfr_normals_fwd: grad_normals (B,3,H,W) f32 -> grad_depth (B,H,W) f32 += (atomics). */
int gcfr_normals_bwd(const float *grad_normals, const float *depth, int32_t B, int32_t H, int32_t W,
                     double fx, double fy, double cx, double cy, float z_offset, int32_t negate_y,
                     float *grad_depth, void *stream);

/*
 * One-call forward for a batch: gcfr_light_prep + depth repack + ray march with the shading fused
 * into the march kernel's epilogue (each lane shades the pixel it just marched; min_dist never
 * makes a round trip through HBM).  Replaces T8:356-522 in one enqueue.  Arguments as in the three
 * entry points above; workspace is mandatory; argmin / shadow_w / full / final_shading may be NULL.
 * Results are bit-identical to calling gcfr_light_prep, gcfr_shadow_fwd and gcfr_shade_fwd in turn.
 */
int gcfr_render_fwd(const float *light_raw, int32_t clamp_z, float clamp_min, float light_distance,
                    const float *depth, const uint8_t *mask_u8, int32_t mask_batch,
                    const float *normals, const float *albedo, const float *ambient, int32_t B,
                    int32_t L, int32_t H, int32_t W, int32_t N, const double *t_table, float bonus,
                    const float *bonus_box, float intensity, float *unit_out, float *light_pt_out,
                    float *min_dist, int32_t *argmin, float *shadow_w, float *full,
                    float *final_shading, float *rendered, void *workspace, size_t workspace_bytes,
                    void *stream, const gcfr_options *opt);

/*
 * gcfr_render_fwd with the normals stage fused in: the march epilogue evaluates the 3x3 depth stencil of
 * gcfr_normals_fwd itself (same device function, same bits), so T8:353-522 is two launches and the
 * (B,3,H,W) normals tensor makes no HBM round trip.  normals_out (B,3,H,W) may be NULL.
 */
int gcfr_render_from_depth_fwd(const float *light_raw, int32_t clamp_z, float clamp_min,
                               float light_distance, const float *depth, const uint8_t *mask_u8,
                               int32_t mask_batch, double fx, double fy, double cx, double cy,
                               float z_offset, int32_t negate_y, const float *albedo, const float *ambient,
                               int32_t B, int32_t L, int32_t H, int32_t W, int32_t N, const double *t_table,
                               float bonus, const float *bonus_box, float intensity, float *unit_out,
                               float *light_pt_out, float *min_dist, int32_t *argmin, float *normals_out,
                               float *shadow_w, float *full, float *final_shading, float *rendered,
                               void *workspace, size_t workspace_bytes, void *stream,
                               const gcfr_options *opt);

/* ---------------------------------------------------------------------------------------------
 * Backward.  The reference has no explicit backward: torch autograd replays T8:352-524
 * (loss.backward() at T8:655).  These entry points compute the same vector-Jacobian products.
 * Buffers marked "+=" are ACCUMULATED into (zero them first); "=" are overwritten.
 * ------------------------------------------------------------------------------------------- */

/*
 * Backward of gcfr_shadow_fwd (T8:371-515 under autograd): only the argmin sample of each pixel
 * carries gradient (torch.min, T8:514); round/floor/ceil, the end-point branch and the clamp are
 * piecewise constant.
 *   grad_min_dist (B,L,H,W) f32   dLoss/d minimum_distance
 *   argmin        (B,L,H,W) i32   from gcfr_shadow_fwd (-1 = masked minimum: no gradient)
 *   grad_depth    (B,H,W) f32 +=  to the four bilinear corners and the pixel's own depth (atomics)
 *   grad_light_pt (B,L,3) f64 +=  to incident_light_points (reduced over pixels in f64)
 */
int gcfr_shadow_bwd(const float *grad_min_dist, const float *depth, const float *light_pt,
                    const int32_t *argmin, int32_t B, int32_t L, int32_t H, int32_t W, int32_t N,
                    const double *t_table, float *grad_depth, double *grad_light_pt, void *stream);

/*
 * Backward of gcfr_shade_fwd (T8:364-369, 517-522 under autograd).
 *   g_shadow_w, g_full, g_final (B,L,H,W) f32, g_rendered (B,L,3,H,W) f32: upstream grads, any may be NULL
 *   grad_normals (B,3,H,W) f32 =   (w.r.t. the un-normalised normals passed to the forward)
 *   grad_albedo  (B,3,H,W) f32 =
 *   grad_depth   (B,H,W) f32 +=    through the incident-light direction only
 *   grad_light_pt (B,L,3) f64 +=,  grad_ambient (B,L) f64 +=
 *   grad_min_dist (B,L,H,W) f32 =  feed to gcfr_shadow_bwd
 */
int gcfr_shade_bwd(const float *normals, const float *depth, const float *albedo,
                   const float *light_pt, const float *ambient, const float *min_dist, int32_t B,
                   int32_t L, int32_t H, int32_t W, float intensity, const float *g_shadow_w,
                   const float *g_full, const float *g_final, const float *g_rendered,
                   float *grad_normals, float *grad_albedo, float *grad_depth, double *grad_light_pt,
                   double *grad_ambient, float *grad_min_dist, void *stream);

/*
 * Backward of gcfr_render_from_depth_fwd in ONE launch: per (image, pixel) the shading backward of every light,
 * the ray-march backward through each light's argmin sample and the normals-stencil backward, with no
 * intermediate grad_min_dist / grad_normals tensors (same per-pixel device functions as the three kernels
 * above, so the same numbers up to atomic ordering).  Follow with gcfr_light_prep_bwd.
 *   normals_fwd (B,3,H,W) f32 or NULL: the unit normals the forward wrote (gcfr_render_from_depth_fwd's normals_out);
 *                 given them the backward evaluates the depth stencil once instead of twice
 *   g_normals_out (B,3,H,W) f32 or NULL: upstream gradient on the returned unit normals
 *   grad_albedo (B,3,H,W) f32 =;  grad_depth (B,H,W) f32 +=;  grad_light_pt (B,L,3) f64 +=;  grad_ambient (B,L) f64 +=
 */
int gcfr_render_bwd(const float *depth, const float *albedo, const float *light_pt, const float *ambient,
                    const float *min_dist, const int32_t *argmin, const float *normals_fwd, int32_t B, int32_t L,
                    int32_t H, int32_t W,
                    int32_t N, const double *t_table, double fx, double fy, double cx, double cy,
                    float z_offset, int32_t negate_y, float intensity, const float *g_shadow_w,
                    const float *g_full, const float *g_final, const float *g_rendered,
                    const float *g_normals_out, float *grad_albedo, float *grad_depth,
                    double *grad_light_pt, double *grad_ambient, void *stream);

/*
 * Backward of gcfr_light_prep (T8:357-363 under autograd).
 *   grad_unit (n,3) f32 or NULL: dLoss/d unit_light_direction (T8:636 uses it in the cosine loss)
 *   grad_light_pt (n,3) f64 or NULL: accumulated by the two kernels above
 *   grad_light_raw (n,3) f32 =
 */
int gcfr_light_prep_bwd(const float *light_raw, int32_t n, int32_t clamp_z, float clamp_min,
                        float light_distance, const float *grad_unit, const double *grad_light_pt,
                        float *grad_light_raw, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Inference-side consumers of the block's outputs: what the test scripts do between the forward
 * and cv2.imwrite, and the MATLAB border fix -- on the device, so a relit batch leaves it as bytes.
 * ------------------------------------------------------------------------------------------- */

/*
 * The images the inference scripts write per face, quantised as cv2.imwrite quantises a float image
 * (round half to even, clip to [0, 255]); RGB, HWC -- the bytes that end up in the PNG.
 * Replaces test_relight_single_image.py:601-620 (rendered image only) and
 * test_raytracing_relighting_CelebAHQ_DSSIM_8x.py:583-608 / ..._lighting_transfer.py:560-579 (all six).
 * B photographs, L relit images per photograph (L = 1: the scripts' one target light per forward, S1:582-588; L > 1: the
 * many-lights forward, all L composites of a face from ONE network pass -- the scripts re-run the model per light):
 *   input_hwc     (B,H,W,3) f32 in [0,1]   the photograph, as the scripts hold it (training_images)
 *   rendered      (B,L,3,H,W) f32          rendered_images                   -> out_rendered (B,L,H,W,3): the relit face
 *                                          pasted into the photograph where mask > 0
 *   (albedo, depth, normals and their outputs are per photograph, (B,...); shadow_w / final_shading and theirs (B,L,H,W))
 *   albedo        (B,3,H,W) f32 or NULL    -> out_albedo  (B,H,W,3) = 255 albedo mask
 *   depth         (B,H,W) f32 or NULL      -> out_depth   (B,H,W)   = 255 (-depth - lo)/(hi - lo) mask, with
 *   depth_range   DEVICE {lo, hi} f32      = min / max of -depth over the whole batch (S8:589-590)
 *   shadow_w, final_shading (B,H,W) f32 or NULL -> out_shadow, out_shading (B,H,W) = 255 x mask
 *   normals       (B,3,H,W) f32 or NULL    -> out_normals (B,H,W,3) = 255 (n + 1)/2 mask
 *   mask          (MB,H,W) u8               the skin mask as read from disk; the kernel forms the scripts' mask/255.0
 *                                          itself; MB = 1 or B
 *   mask_f32      0: the mask is f64 as in test_relight_single_image.py:580 / S8:569-578 (numpy f64 array / 255.0);
 *                 1: f32 as in test_relight_single_image_lighting_transfer.py:540 (torch uint8 tensor / 255.0), which
 *                    keeps the shadow-mask and depth images' products in f32 (SLT:575, 577)
 * final_shading and normals are f64 in the reference; they are widened to f64 before the arithmetic.
 * Pinned to the reference's own main() (tests/golden/slt_main_*.npz).  Any out_* except out_rendered may be NULL.
 */
int gcfr_inference_images_u8(const float *input_hwc, const float *rendered, const float *albedo, const float *depth,
                             const float *depth_range, const float *shadow_w, const float *final_shading,
                             const float *normals, const uint8_t *mask, int32_t mask_batch, int32_t B, int32_t L,
                             int32_t H, int32_t W, uint8_t *out_rendered, uint8_t *out_shadow, uint8_t *out_albedo,
                             uint8_t *out_depth, uint8_t *out_shading, uint8_t *out_normals, int32_t mask_f32,
                             void *stream);

/*
 * fix_border_artifacts_CVPR2022.m:1-18: pixels on the border of the face mask (0 < 7x7 box sum of the rounded
 * mask < 30, zero padding) take the 3x3 median (zero padding) of their channel.
 *   img_hwc (B,H,W,3) u8;  face_mask_u8 (MB,H,W) u8 skin mask as stored on disk (0 ... 255);  out_hwc (B,H,W,3) u8,
 *   must not alias img_hwc.
 */
int gcfr_fix_border_u8(const uint8_t *img_hwc, const uint8_t *face_mask_u8, int32_t mask_batch, int32_t B, int32_t H,
                       int32_t W, uint8_t *out_hwc, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Data formats either side of the block: batch assembly from the bytes load_data() reads, and the
 * MATLAB evaluation scripts' two metrics as device reductions.
 * ------------------------------------------------------------------------------------------- */

/*
 * One training batch from the uint8 arrays as stored on disk -- what
 * train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:545-556 (load_data) and :607-615 (batch slicing) do in float64 on
 * the host for the whole dataset, here per batch on the device:
 *   images_u8     (B,H,W,3) u8  imread(jpg)                      -> images     (B,H,W,3) f32 = f32(u8 / 255.0)      T8:550, 618
 *   depth_mask_u8 (B,H,W)   u8  imread(depth mask)               -> masks      (B,H,W)   f32 = u8 / 255.0           T8:546, 610
 *   face_mask_u8  (B,H,W)   u8  imread(face mask)                -> masks_fill (B,H,W)   f32 = (max(face, depth) > 128) ? 1 : 0   T8:552-556, 612
 *   albedo_u8     (B,H,W)   u8  imread(grey albedo)              -> albedo     (B,H,W)   f32 = u8 / 255.0           T8:551, 615
 * masks / masks_fill (with face_mask_u8) / albedo (with albedo_u8) may be NULL.
 */
int gcfr_assemble_batch_u8(const uint8_t *images_u8, const uint8_t *depth_mask_u8, const uint8_t *face_mask_u8,
                           const uint8_t *albedo_u8, int32_t B, int32_t H, int32_t W, float *images, float *masks,
                           float *masks_fill, float *albedo, void *stream);

/*
 * Masked MSE (MSE_MP.m:24) and masked DSSIM (DSSIM_MP_RGB.m:24-26) of B image pairs, f64:
 *   recon_u8, gt_u8 (B,H,W,3) u8 RGB;  mask_u8 (MB,H,W) u8 (MB = 1 or B), all scaled by 1/255.0 as the scripts do;
 *   mse_out, dssim_out (B) f64, either may be NULL;
 *   workspace: gcfr_masked_metrics_workspace_bytes(B,H,W) bytes of device scratch, 8-byte aligned.
 * DSSIM follows MATLAB ssim()'s documented defaults on an M x N x 3 volume (Gaussian sigma 1.5, radius 5, replicate padding
 * on all three axes, K = (0.01, 0.03), dynamic range 1).  PARITY UNPINNED: MATLAB is not available to the builder.
 */
size_t gcfr_masked_metrics_workspace_bytes(int32_t B, int32_t H, int32_t W);
int gcfr_masked_metrics_u8(const uint8_t *recon_u8, const uint8_t *gt_u8, const uint8_t *mask_u8, int32_t mask_batch,
                           int32_t B, int32_t H, int32_t W, double *mse_out, double *dssim_out, void *workspace,
                           size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement aid (bench.py `roofline.hbm_measured_copy_GBs`): a float4 grid-stride device-to-device copy of `bytes` bytes
 * (multiple of 16, both pointers 16-byte aligned), one workgroup of 256 lanes per CU, four loads in flight per lane, non-temporal --
 * the achievable-HBM probe (6.3 TB/s, read + write) the roofline's 8 TB/s spec peak is reported beside.  Not part of the render path.
 * ------------------------------------------------------------------------------------------- */
int gcfr_copy_probe(const void *src, void *dst, size_t bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* GCFR_H */
