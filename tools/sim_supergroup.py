"""Analysis only (CPU, numpy f64 statistics -- not a parity tool): what a SECOND level of the depth-bound skip could save.

Level 1 is what the march does: per group of G = 4 samples a band test against the stride-8 plane tile (tools/sim_plane.py).
Level 2 tests a super-group of SG = 16 samples against a stride-32 plane tile (region 64 x 64) BEFORE any mask byte of the
super-group is read, and skips all four groups if no lane of the wave can win (and every lane already holds a minimum).
Counts per face, for 16 x 4 wave tiles: groups visited / bodies executed by the current scheme (mask bounding-box range + early
termination against the image's depth maximum), and super-group tests / groups still visited with level 2; prices them with
the instruction costs measured on the kernel (49 bookkeeping + 35 test per visited group, 216 per body, ~70 per super-group test).
"""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/oracle")
import bench  # noqa: E402
import c_oracle  # noqa: E402

C_VISIT, C_TEST, C_BODY, C_SG = 49.0, 35.0, 216.0, 70.0


def plane_tiles(depth, stride, H, W):
    nth, ntw = H // stride + 1, W // stride + 1
    dext = np.vstack([depth[-1:], depth])
    dext = np.hstack([dext[:, -1:], dext])
    ce, re = np.arange(W + 1) - 1.0, np.arange(H + 1) - 1.0
    XE, YE = np.meshgrid(ce - W / 2.0, H / 2.0 - re)
    pa, pb, clo, chi = (np.zeros((nth, ntw)) for _ in range(4))
    for i in range(nth):
        for j in range(ntw):
            sl = (slice(i * stride, i * stride + 2 * stride), slice(j * stride, j * stride + 2 * stride))
            z, X, Y = dext[sl], XE[sl], YE[sl]
            A = np.stack([X.ravel(), Y.ravel(), np.ones(X.size)], 1)
            coef, *_ = np.linalg.lstsq(A, z.ravel(), rcond=None)
            a, b = np.clip(coef[0], -4, 4), np.clip(coef[1], -4, 4)
            res = z - (a * X + b * Y)
            pa[i, j], pb[i, j], clo[i, j], chi[i, j] = a, b, res.min(), res.max()
    return pa, pb, clo, chi


def run(seed=0, tile=(4, 16), G=4, SG=16, H=256, W=256, N=160, t0=0.025, dt=0.005, data="synthetic"):
    if data == "ffhq":
        depth, mask, _a, _n, light, _amb = bench.ffhq_faces(1, seed)
    else:
        depth, mask, _a, _n, light, _amb = bench.synth_faces(1, seed)
    depth, mask = depth[0].astype(np.float64), mask[0]
    _, pt = c_oracle.light_prep(light, clamp_z_min=0.0)
    Cx, Cy, Cz = [float(v) for v in pt[0]]
    rr, cc = np.mgrid[0:H, 0:W]
    x, y = cc - W / 2.0, H / 2.0 - rr
    ux, uy = Cx - x, Cy - y
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(ux > 0, (W / 2.0 - 1 - x) / ux, np.where(ux < 0, (-W / 2.0 - x) / ux, np.inf))
        ty = np.where(uy > 0, (H / 2.0 - y) / uy, np.where(uy < 0, (-H / 2.0 + 1 - y) / uy, np.inf))
    te = np.minimum(np.minimum(tx, ty), 1.0)
    dx, dy = te * ux, te * uy
    zb = depth
    BCz = Cz - zb
    n = np.sqrt(ux * ux + uy * uy)
    proj = (dx * ux + dy * uy) / np.maximum(n, 1e-9)
    c1 = BCz * proj
    t = t0 + dt * np.arange(N)
    err = 4e-3 * np.abs(BCz) + 1e-6 * np.abs(c1) * t[-1] + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * 400 \
        + n * (1.2e-2 + 8e-6 * max(H, W))
    L1 = plane_tiles(depth, 8, H, W)
    L2 = plane_tiles(depth, 32, H, W)
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    wall = lambda a: a.reshape(wsh).all(axis=(1, 3))
    wany = lambda a: a.reshape(wsh).any(axis=(1, 3))
    lanes = lambda w: np.repeat(np.repeat(w, th, 0), tw, 1)
    # mask bounding box -> per-lane / per-wave sample ranges
    rows, cols = np.nonzero(mask)
    X0, X1 = cols.min() - W / 2.0 - 0.51, cols.max() - W / 2.0 + 0.51
    Y0, Y1 = H / 2.0 - rows.max() - 0.51, H / 2.0 - rows.min() + 0.51
    inb = np.zeros((N, H, W), bool)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        inb[k] = (sx >= X0) & (sx <= X1) & (sy >= Y0) & (sy <= Y1)
    lane_lo = np.where(inb.any(0), inb.argmax(0), N)
    lane_hi = np.where(inb.any(0), N - 1 - inb[::-1].argmax(0), -1)
    w_lo, w_hi = lane_lo.reshape(wsh).min(axis=(1, 3)), lane_hi.reshape(wsh).max(axis=(1, 3))
    gz_hi = max(depth.max(), 0.0)

    def band_cannot_win(tiles_, stride, ka, kb, best):
        pa, pb, clo, chi = tiles_
        ca = np.rint(x + t[ka] * dx).astype(int) + W // 2
        cb = np.rint(x + t[kb] * dx).astype(int) + W // 2
        ra = H // 2 - np.rint(y + t[ka] * dy).astype(int)
        rb = H // 2 - np.rint(y + t[kb] * dy).astype(int)
        tj, ti = np.clip(np.minimum(ca, cb), 0, W - 1) // stride, np.clip(np.minimum(ra, rb), 0, H - 1) // stride
        a, b = pa[ti, tj], pb[ti, tj]

        def Gend(tk, c):
            return n * (a * (x + tk * dx) + b * (y + tk * dy) + c - zb) - c1 * tk
        lo = np.minimum(Gend(t[ka], clo[ti, tj]), Gend(t[kb], clo[ti, tj]))
        hi = np.maximum(Gend(t[ka], chi[ti, tj]), Gend(t[kb], chi[ti, tj]))
        gp = np.maximum(lo, -hi)
        Ta, Tb = c1 * t[ka], c1 * t[kb]
        gap0 = np.maximum(-n * zb - np.maximum(Ta, Tb), np.minimum(Ta, Tb) + n * zb)
        g = np.minimum(gp, gap0) - err
        return (g > 0) & (g * g * 0.998 > best)

    best = np.full((H, W), np.inf)
    alive = np.ones(w_lo.shape, bool)
    visits = tests = bodies = 0
    sg_tests = sg_skipped = visits2 = tests2 = 0
    sg_skip_now = np.zeros(w_lo.shape, bool)
    viol = 0
    for g0 in range(0, N, G):
        ks = np.arange(g0, min(N, g0 + G))
        in_range = (w_lo <= ks[-1]) & (w_hi >= ks[0]) & alive
        if g0 % SG == 0:      # super-group boundary: level-2 test for the waves that would visit its first group
            ke = min(N, g0 + SG) - 1
            cw2 = band_cannot_win(L2, 32, g0, ke, best) & np.isfinite(best)
            sg_cand = (w_lo <= ke) & (w_hi >= g0) & alive
            sg_skip_now = sg_cand & wall(cw2)
            sg_tests += int(sg_cand.sum())
            sg_skipped += int(sg_skip_now.sum())
        # the samples of this group
        Sg = np.full((H, W), np.inf)
        anyun = np.zeros((H, W), bool)
        for k in ks:
            sx, sy = x + t[k] * dx, y + t[k] * dy
            col = np.rint(sx).astype(int) + W // 2
            row = H // 2 - np.rint(sy).astype(int)
            m = mask[np.clip(row, 0, H - 1), np.clip(col, 0, W - 1)] != 0
            u, v = sx + W / 2.0 - 1e-4, H / 2.0 - sy - 1e-4
            fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
            cu, cv = np.clip(fu + 1, 0, W - 1), np.clip(fv + 1, 0, H - 1)
            wx1, wy1 = u - fu, v - fv
            z = (depth[fv, fu] * (1 - wx1) + depth[fv, cu] * wx1) * (1 - wy1) + (depth[cv, fu] * (1 - wx1) + depth[cv, cu] * wx1) * wy1
            BAx, BAy, BAz = sx - 1e-4 - x, sy + 1e-4 - y, z - zb
            S = (BAy * BCz - BAz * uy) ** 2 + (BAz * ux - BAx * BCz) ** 2 + (BAx * uy - BAy * ux) ** 2
            Sg = np.minimum(Sg, np.where(m, S, np.inf))
            anyun |= m
        cw1 = band_cannot_win(L1, 8, ks[0], ks[-1], best)
        some_unmasked = wany(anyun) & in_range
        run_body = wany(anyun & ~cw1) & in_range
        visits += int(in_range.sum())
        tests += int(some_unmasked.sum())
        bodies += int(run_body.sum())
        v2 = in_range & ~sg_skip_now
        visits2 += int(v2.sum())
        tests2 += int((some_unmasked & ~sg_skip_now).sum())
        viol += int((run_body & sg_skip_now).sum())      # (this model's error terms are the kernel's, its arithmetic is not: a handful of borderline cases)
        best = np.where(lanes(run_body), np.minimum(best, Sg), best)
        knext = ks[-1] + 1
        if knext < N and (g0 // G) % 2 == 1:               # early termination, every other group
            gd = c1 * t[knext] - n * (gz_hi - zb) - err
            done = ((c1 > 0) & (gd > 0) & (gd * gd * 0.998 > best)) | (lane_hi < knext)
            alive &= ~wall(done)
    base = visits * C_VISIT + tests * C_TEST + bodies * C_BODY
    hier = sg_tests * C_SG + visits2 * C_VISIT + tests2 * C_TEST + bodies * C_BODY
    if viol:
        print("  (%d group-visits of skipped super-groups would have run a body in this f64 model)" % viol)
    print("%s seed %d: visits %d tests %d bodies %d | super-groups tested %d skipped %d (%.0f %%) -> visits %d tests %d | loop cost %.2f M -> %.2f M (%.1f %%)"
          % (data, seed, visits, tests, bodies, sg_tests, sg_skipped, 100.0 * sg_skipped / max(sg_tests, 1), visits2, tests2,
             base / 1e6, hier / 1e6, 100.0 * (hier - base) / base))
    return base, hier


if __name__ == "__main__":
    tot = np.zeros(2)
    for d in ("synthetic", "ffhq"):
        for s in range(3):
            tot += run(seed=s, data=d)
    print("total loop cost %.2f M -> %.2f M (%.1f %%)" % (tot[0] / 1e6, tot[1] / 1e6, 100.0 * (tot[1] - tot[0]) / tot[0]))
