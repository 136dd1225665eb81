"""Analysis only: plane-fit depth bounds per tile (z in a*x + b*y + [c_lo, c_hi]) vs min/max bounds."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench, c_oracle


def run(seed=0, tile=(8, 8), G=4, stride=8, H=256, W=256, N=160, t0=0.025, dt=0.005, light_override=None):
    depth, mask, *_rest = bench.synth_faces(1, seed)
    light = _rest[2]
    if light_override is not None:
        light = np.array([light_override], np.float32)
    depth, mask = depth[0].astype(np.float64), mask[0]
    _, pt = c_oracle.light_prep(light, clamp_z_min=0.0)
    Cx, Cy, Cz = [float(v) for v in pt[0]]
    rr, cc = np.mgrid[0:H, 0:W]
    x = cc - W / 2.0; y = H / 2.0 - rr
    ux, uy = Cx - x, Cy - y
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(ux > 0, (W / 2.0 - 1 - x) / ux, np.where(ux < 0, (-W / 2.0 - x) / ux, np.inf))
        ty = np.where(uy > 0, (H / 2.0 - y) / uy, np.where(uy < 0, (-H / 2.0 + 1 - y) / uy, np.inf))
    te = np.minimum(np.minimum(tx, ty), 1.0)
    dx, dy = te * ux, te * uy
    zb = depth; BCz = Cz - zb
    n = np.sqrt(ux * ux + uy * uy)
    proj = (dx * ux + dy * uy) / np.maximum(n, 1e-9)
    t = t0 + dt * np.arange(N)
    nth, ntw = H // stride + 1, W // stride + 1
    dext = np.vstack([depth[-1:], depth]); dext = np.hstack([dext[:, -1:], dext])
    # coordinates of extended cells in the kernel's (x, y) frame: col c -> x = c - W/2 ; row r -> y = H/2 - r
    ce = np.arange(W + 1) - 1.0; re = np.arange(H + 1) - 1.0
    XE, YE = np.meshgrid(ce - W / 2.0, H / 2.0 - re)
    # wrap row/col make the plane model useless there; treat by large residuals naturally
    pa = np.zeros((nth, ntw)); pb = np.zeros((nth, ntw)); clo = np.zeros((nth, ntw)); chi = np.zeros((nth, ntw))
    zmn = np.zeros((nth, ntw)); zmx = np.zeros((nth, ntw))
    for i in range(nth):
        for j in range(ntw):
            sl = (slice(i * stride, i * stride + 2 * stride), slice(j * stride, j * stride + 2 * stride))
            z, X, Y = dext[sl], XE[sl], YE[sl]
            A = np.stack([X.ravel(), Y.ravel(), np.ones(X.size)], 1)
            coef, *_ = np.linalg.lstsq(A, z.ravel(), rcond=None)
            a, b = np.clip(coef[0], -16, 16), np.clip(coef[1], -16, 16)
            res = z - (a * X + b * Y)
            pa[i, j], pb[i, j], clo[i, j], chi[i, j] = a, b, res.min(), res.max()
            zmn[i, j], zmx[i, j] = z.min(), z.max()
    best = np.full((H, W), np.inf)
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    wany = lambda a: a.reshape(wsh).any(axis=(1, 3))
    n_mask = n_mm = n_pl = n_perf = 0
    for g in range(0, N, G):
        ks = np.arange(g, min(N, g + G))
        Sg = np.full((H, W), np.inf); anyun = np.zeros((H, W), bool)
        cols = []; rows = []
        for k in ks:
            sx, sy = x + t[k] * dx, y + t[k] * dy
            col = np.rint(sx).astype(int) + W // 2; row = H // 2 - np.rint(sy).astype(int)
            cols.append(col); rows.append(row)
            m = mask[np.clip(row, 0, H - 1), np.clip(col, 0, W - 1)] != 0
            u, v = sx + W / 2.0 - 1e-4, H / 2.0 - sy - 1e-4
            fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
            cu, cv = np.clip(fu + 1, 0, W - 1), np.clip(fv + 1, 0, H - 1)
            wx1, wy1 = u - fu, v - fv
            z = (depth[fv, fu] * (1 - wx1) + depth[fv, cu] * wx1) * (1 - wy1) + (depth[cv, fu] * (1 - wx1) + depth[cv, cu] * wx1) * wy1
            BAx, BAy, BAz = sx - 1e-4 - x, sy + 1e-4 - y, z - zb
            S = (BAy * BCz - BAz * uy) ** 2 + (BAz * ux - BAx * BCz) ** 2 + (BAx * uy - BAy * ux) ** 2
            Sg = np.minimum(Sg, np.where(m, S, np.inf)); anyun |= m
        cmin = np.minimum(cols[0], cols[-1]); cmax = np.maximum(cols[0], cols[-1])
        rmin = np.minimum(rows[0], rows[-1]); rmax = np.maximum(rows[0], rows[-1])
        tj, ti = np.maximum(cmin, 0) // stride, np.maximum(rmin, 0) // stride
        cov = (cmin >= 0) & (rmin >= 0) & (cmax <= W - 1) & (rmax <= H - 1) & (cmax + 2 <= (tj + 2) * stride - 1) & (rmax + 2 <= (ti + 2) * stride - 1)
        ta, tb = t[ks[0]], t[ks[-1]]
        Ta, Tb = BCz * ta * proj, BCz * tb * proj
        err = 4e-3 * np.abs(BCz) + 1e-6 * np.abs(BCz * proj) * t[-1] + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * 400
        thr = 1.001 * np.sqrt(best)
        # min/max
        Tlo, Thi = np.minimum(Ta, Tb), np.maximum(Ta, Tb)
        gap = np.maximum(n * (zmn[ti, tj] - zb) - Thi, Tlo - n * (zmx[ti, tj] - zb))
        gap0 = np.maximum(-n * zb - Thi, Tlo + n * zb)
        skip_mm = cov & (np.minimum(gap, gap0) - err > thr)
        # plane: z(s_k) in plane(s_k) + [clo, chi]; s_k = (x + t dx - 1e-4.., y + t dy ..)
        a, b = pa[ti, tj], pb[ti, tj]
        def Gend(tk, c):
            return n * (a * (x + tk * dx) + b * (y + tk * dy) + c - zb) - BCz * tk * proj
        lo_a, lo_b = Gend(ta, clo[ti, tj]), Gend(tb, clo[ti, tj])
        hi_a, hi_b = Gend(ta, chi[ti, tj]), Gend(tb, chi[ti, tj])
        gp = np.maximum(np.minimum(lo_a, lo_b), -np.maximum(hi_a, hi_b))   # >0: surface band entirely above or below the ray
        errp = err + 1e-6 * n * 16 * 2 * max(H, W) + n * (np.abs(a) + np.abs(b)) * 1e-3
        skip_pl = cov & (np.minimum(gp, gap0) - errp > thr)
        skip_pl |= skip_mm
        assert not (skip_pl & (Sg < best)).any(), int((skip_pl & (Sg < best)).sum())
        n_mask += wany(anyun).sum(); n_mm += wany(anyun & ~skip_mm).sum(); n_pl += wany(anyun & ~skip_pl).sum()
        n_perf += wany(Sg < best).sum()
        best = np.minimum(best, Sg)
    print(f"seed {seed}: mask {n_mask} minmax {n_mm} plane {n_pl} perfect {n_perf}")


if __name__ == '__main__':
    for s in range(3):
        run(seed=s)
