/*
 * TEST INFRASTRUCTURE -- CPU oracle for the GeomConsistentFR render block.
 *
 * A scalar, per-pixel restatement of the reference's ray-marched soft shadow + Lambertian
 * shading/compositing block (train_raytracing_relighting_CelebAHQ_DSSIM_8x.py:352-524, "T8"),
 * written from the arithmetic spec in SURVEY.md Appendix A.  Each function cites the reference
 * lines it follows.  The reference materialises (N,2,H,W) tensors and loops over images; this
 * file loops over pixels and samples and keeps the reference's dtypes and *operation order*:
 *   - line parameters / end points: f32, every mul/add/div separately rounded
 *   - sample positions, bilinear weights, interpolated depth: f64, separately rounded
 *   - point-to-line distance: f32
 * Build with -ffp-contract=off (see oracle/Makefile) so the compiler never fuses a*b+c.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the reported CPU baseline -- never as a product path.
 *
 * Pinning: validated against the reference itself (imported through oracle/ref_shim.py) and
 * against the committed golden vectors in tests/golden/ (tests/test_oracle_golden.py).
 * Normals are NOT computed here; they are an input (kornia 0.4.1 is un-vendored: parity unpinned,
 * see oracle/normals_restatement.py).
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

/* np.arange(t0, stop, dt) value rule: first + k*(second-first), all f64 (T8:468). */
void gcfr_oracle_sample_table(double t0, double dt, int n, double *out)
{
    double delta = (t0 + dt) - t0;
    for (int k = 0; k < n; ++k)
        out[k] = t0 + (double)k * delta;
}

/*
 * Light preparation, T8:357-363 (training: clamp z at `clamp_min`) / S1:332-336 (inference:
 * no clamp).  l = (a, b, max(c, clamp_min)); u = l / max(||l||_2, 1e-12); C = light_distance*u.
 * light_raw: (B,3) f32.  unit_out, light_pt_out: (B,3) f32.
 */
void gcfr_oracle_light_prep(const float *light_raw, int B, int clamp_z, float clamp_min,
                            float light_distance, float *unit_out, float *light_pt_out)
{
    for (int b = 0; b < B; ++b) {
        float a = light_raw[3 * b + 0], bb = light_raw[3 * b + 1], c = light_raw[3 * b + 2];
        if (clamp_z)
            c = (c > clamp_min) ? c : clamp_min; /* torch.maximum(l_z, 0) T8:358 */
        /* F.normalize T8:360.  torch's 2-norm reduction accumulates acc = fma(x, x, acc) in f32
         * (probed on torch 2.10 CPU: bit-equal on 1e5 random vectors; plain (a*a+b*b)+c*c is not). */
        float n = sqrtf(fmaf(c, c, fmaf(bb, bb, a * a)));
        float d = (n > 1e-12f) ? n : 1e-12f;
        float ux = a / d, uy = bb / d, uz = c / d;
        unit_out[3 * b + 0] = ux;
        unit_out[3 * b + 1] = uy;
        unit_out[3 * b + 2] = uz;
        light_pt_out[3 * b + 0] = light_distance * ux; /* T8:362 */
        light_pt_out[3 * b + 1] = light_distance * uy;
        light_pt_out[3 * b + 2] = light_distance * uz;
    }
}

/* End point of the pixel->light 2-D segment clipped to the image box. T8:378-465. */
static void end_point(float x, float y, float Cx, float Cy, int H, int W, float *Ex, float *Ey)
{
    const double x_lo = -(W / 2.0), x_hi = (W - W / 2.0 - 1.0); /* T8:386, 416 */
    const double y_lo = 1.0 - H / 2.0, y_hi = H / 2.0;          /* T8:387, 399 */
    const float e4 = 0.0001f;

    float m = (Cy - y) / ((Cx - x) + e4); /* slopes     T8:378 */
    float ic = Cy - m * Cx;               /* intercepts T8:379 */
    double LX = (double)Cx, LY = (double)Cy; /* T8:380-381 */

    int xcase = (LX < x_lo) ? 0 : (LX <= x_hi ? 1 : 2);
    int ycase = (LY < y_lo) ? 0 : (LY <= y_hi ? 1 : 2);
    float ex, ey;
    if (xcase == 1) {
        if (ycase == 1) { /* T8:422-425: the light's own xy */
            ex = Cx;
            ey = Cy;
        } else { /* T8:417-421 / 426-430 */
            float yb = (float)(ycase == 0 ? y_lo : y_hi);
            ex = (yb - ic) / (m + e4);
            ey = yb;
        }
    } else {
        float xb = (float)(xcase == 0 ? x_lo : x_hi);
        float Xx = xb, Xy = m * xb + ic; /* "try x = ..." T8:389-390 */
        if (ycase == 1) {                /* T8:399-403 / 444-448 */
            ex = Xx;
            ey = Xy;
        } else { /* T8:387-398, 404-415, 432-443, 449-460 */
            float yb = (float)(ycase == 0 ? y_lo : y_hi);
            float Yx = (yb - ic) / (m + e4), Yy = yb;
            float b = (Yx >= (float)x_lo && Yx <= (float)x_hi) ? 1.0f : 0.0f; /* intersects_y */
            float nb = 1.0f - b;
            ex = Yx * b + Xx * nb; /* arithmetic select, T8:398 */
            ey = Yy * b + Xy * nb;
        }
    }
    /* clamp T8:462-465 (constants -128,127,-127,128 generalised to the box) */
    if (ex < (float)x_lo) ex = (float)x_lo;
    if (ex > (float)x_hi) ex = (float)x_hi;
    if (ey < (float)y_lo) ey = (float)y_lo;
    if (ey > (float)y_hi) ey = (float)y_hi;
    *Ex = ex;
    *Ey = ey;
}

static inline int wrap(int i, int n) { return i < 0 ? i + n : i; } /* negative index wraps, T8:488-491 */

/*
 * Ray-marched minimum point-to-line distance, T8:374-515, for B images x L lights.
 *   depth      (B,H,W) f32      c2_o_depth (already x100, T8:350)
 *   mask_u8    (MB,H,W) u8      1 where the reference's mask != 0; MB = B (T8:510) or 1 (S1:488)
 *   light_pt   (B,L,3) f32      incident_light_points = light_distance * unit dir (T8:362)
 *   t_table    (N) f64          sample fractions
 *   bonus      added to the minimum if the light's xy lies in [bx_lo,bx_hi]x[by_lo,by_hi]
 *              (S1:495-496, SLT:503-504); pass bonus = 0 for the training form.
 *   min_dist   (B,L,H,W) f32 out;  argmin (B,L,H,W) i32 out (first index of the minimum)
 */
void gcfr_oracle_shadow_min_distance(const float *depth, const uint8_t *mask_u8, int mask_batch,
                                     const float *light_pt, int B, int L, int H, int W, int N,
                                     const double *t_table, float bonus, float bx_lo, float bx_hi,
                                     float by_lo, float by_hi, float *min_dist, int32_t *argmin)
{
    const double halfW = W / 2.0, halfH = H / 2.0;
    const size_t P = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int bl = 0; bl < B * L; ++bl) {
        for (int r = 0; r < H; ++r) {
            const int b = bl / L;
            const float *z = depth + (size_t)b * P;
            const uint8_t *mk = mask_u8 + (size_t)(mask_batch == 1 ? 0 : b) * P;
            const float Cx = light_pt[3 * bl + 0], Cy = light_pt[3 * bl + 1], Cz = light_pt[3 * bl + 2];
            const int inside = ((double)Cx >= bx_lo && (double)Cx <= bx_hi && (double)Cy >= by_lo &&
                                (double)Cy <= by_hi);
            for (int c = 0; c < W; ++c) {
                /* pixel grids T8:51-55; points_3D T8:356 */
                const float x = (float)c - (float)halfW;
                const float y = (float)halfH - (float)r;
                const float zb = z[(size_t)r * W + c];
                float Ex, Ey;
                end_point(x, y, Cx, Cy, H, W, &Ex, &Ey);
                const float dx = Ex - x, dy = Ey - y; /* difference T8:467 */
                const float BCx = Cx - x, BCy = Cy - y, BCz = Cz - zb; /* T8:507 */
                const float den = sqrtf(((BCx * BCx + BCy * BCy) + BCz * BCz) + 0.0001f);

                float best = INFINITY;
                int besti = 0;
                for (int k = 0; k < N; ++k) {
                    const double t = t_table[k];
                    const double sx = (double)x + t * (double)dx; /* T8:472/480, f64 */
                    const double sy = (double)y + t * (double)dy;
                    /* rounded cell for the mask, T8:472-477 */
                    const int col_r = (int)(rint(sx) + halfW);
                    const int row_r = (int)(halfH - rint(sy));
                    /* unrounded position, T8:480-487 */
                    const double ux = (sx + halfW) - 0.0001;
                    const double uy = (halfH - sy) - 0.0001;
                    const int fx = (int)floor(ux), gx = (int)ceil(ux);
                    const int fy = (int)floor(uy), gy = (int)ceil(uy);
                    const double wx0 = (double)gx - ux, wx1 = ux - (double)fx;
                    const double wy0 = (double)gy - uy, wy1 = uy - (double)fy;
                    const int fxw = wrap(fx, W), gxw = wrap(gx, W), fyw = wrap(fy, H), gyw = wrap(gy, H);
                    const double zUL = z[(size_t)fyw * W + fxw], zUR = z[(size_t)fyw * W + gxw];
                    const double zLL = z[(size_t)gyw * W + fxw], zLR = z[(size_t)gyw * W + gxw];
                    const double up = zUL * wx0 + zUR * wx1;  /* T8:492 */
                    const double low = zLL * wx0 + zLR * wx1; /* T8:493 */
                    const double zA = up * wy0 + low * wy1;   /* T8:494 */
                    /* point A T8:497-502 */
                    const float Ax = (float)(ux - halfW), Ay = (float)(halfH - uy), Az = (float)zA;
                    const float BAx = Ax - x, BAy = Ay - y, BAz = Az - zb; /* T8:504 */
                    /* cross(BA, BC) T8:508 */
                    /* torch's cross kernel is compiled with contraction: a1*b2 - a2*b1 evaluates as
                     * fma(a1, b2, -rnd(a2*b1)) (probed: bit-equal on 1e5 random vectors). */
                    const float Xx = fmaf(BAy, BCz, -(BAz * BCy));
                    const float Xy = fmaf(BAz, BCx, -(BAx * BCz));
                    const float Xz = fmaf(BAx, BCy, -(BAy * BCx));
                    const float num = sqrtf(((Xx * Xx + Xy * Xy) + Xz * Xz) + 0.0001f);
                    float d = num / den; /* T8:509 */
                    /* mask T8:510-512 (arithmetic form) */
                    const int out = (mk[(size_t)wrap(row_r, H) * W + wrap(col_r, W)] == 0);
                    d = (out ? 0.0f : 1.0f) * d + (out ? 1.0f : 0.0f) * 1000000.0f;
                    if (d < best) { /* first minimum, T8:514 */
                        best = d;
                        besti = k;
                    }
                }
                if (inside)
                    best = best + bonus;
                min_dist[(size_t)bl * P + (size_t)r * W + c] = best;
                argmin[(size_t)bl * P + (size_t)r * W + c] = besti;
            }
        }
    }
}

/*
 * Soft-shadow transfer + Lambert shading + composite, T8:364-369, 517-522, given normals.
 *   normals   (B,3,H,W) f64   kornia output with y already negated (T8:353-354), NOT yet re-normalised
 *   depth     (B,H,W) f32;  albedo (B,3,H,W) f32;  light_pt (B,L,3) f32;  ambient (B,L) f32
 *   min_dist  (B,L,H,W) f32
 * outputs: shadow_w (B,L,H,W) f32, full_shading/final_shading (B,L,H,W) f64, rendered (B,L,3,H,W) f32,
 *          normals_out (B,3,H,W) f64 re-normalised (T8:365).
 * dtypes follow torch promotion in the reference: normals f64 (f64 camera matrix), light dir f32.
 */
void gcfr_oracle_shade(const double *normals, const float *depth, const float *albedo,
                       const float *light_pt, const float *ambient, const float *min_dist, int B,
                       int L, int H, int W, float intensity, float *shadow_w, double *full_shading,
                       double *final_shading, float *rendered, double *normals_out)
{
    const size_t P = (size_t)H * W;
    const float halfW = (float)(W / 2.0), halfH = (float)(H / 2.0);
#pragma omp parallel for schedule(static)
    for (int bl = 0; bl < B * L; ++bl) {
        const int b = bl / L;
        const float Cx = light_pt[3 * bl + 0], Cy = light_pt[3 * bl + 1], Cz = light_pt[3 * bl + 2];
        const float amb = ambient[bl];
        for (size_t p = 0; p < P; ++p) {
            const int r = (int)(p / W), c = (int)(p % W);
            const float x = (float)c - halfW, y = halfH - (float)r, zb = depth[b * P + p];
            /* incident_light = normalize(C - P) f32, T8:364 */
            const float lx = Cx - x, ly = Cy - y, lz = Cz - zb;
            float ln = sqrtf(fmaf(lz, lz, fmaf(ly, ly, lx * lx))); /* fma-chain norm, see light_prep */
            ln = ln > 1e-12f ? ln : 1e-12f;
            const float ux = lx / ln, uy = ly / ln, uz = lz / ln;
            /* surface normal re-normalised in f64, T8:365 */
            const double nx = normals[(b * 3 + 0) * P + p], ny = normals[(b * 3 + 1) * P + p],
                         nz = normals[(b * 3 + 2) * P + p];
            double nn = sqrt(fma(nz, nz, fma(ny, ny, nx * nx)));
            nn = nn > 1e-12 ? nn : 1e-12;
            const double n0 = nx / nn, n1 = ny / nn, n2 = nz / nn;
            if (normals_out && bl % L == 0) {
                normals_out[(b * 3 + 0) * P + p] = n0;
                normals_out[(b * 3 + 1) * P + p] = n1;
                normals_out[(b * 3 + 2) * P + p] = n2;
            }
            double dot = (n0 * (double)ux + n1 * (double)uy) + n2 * (double)uz; /* T8:366 */
            double dir = (double)intensity * (dot > 0.0 ? dot : 0.0);
            double full = (double)amb + dir; /* T8:369 */
            /* shadow transfer T8:517 (f32) */
            const float d = min_dist[bl * P + p];
            const float e = expf(-d);
            const float onepe = 1.0f + e;
            const float w = (-4.0f * e) / (onepe * onepe) + 1.0f;
            /* composite T8:518-522 */
            double fin = (double)w * full + (double)(1.0f - w) * (double)amb;
            shadow_w[bl * P + p] = w;
            full_shading[bl * P + p] = full;
            final_shading[bl * P + p] = fin;
            for (int ch = 0; ch < 3; ++ch)
                rendered[((size_t)bl * 3 + ch) * P + p] =
                    (float)((double)albedo[((size_t)b * 3 + ch) * P + p] * fin);
        }
    }
}

int gcfr_oracle_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
