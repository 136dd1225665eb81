"""Analysis only: cost model of a two-level (super-group / group) depth+mask tile skip for the march.
Numpy f64 statistics, not a parity tool.  See tools/sim_depth_bound.py for the single-level version."""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/oracle")
import bench  # noqa: E402
import c_oracle  # noqa: E402

C_TEST, C_BODY, C_PREF = 55.0, 250.0, 62.0


def tiles(depth, mask, stride):
    H, W = depth.shape
    nth, ntw = H // stride + 1, W // stride + 1
    zmin = np.full((nth, ntw), np.inf)
    zmax = np.full((nth, ntw), -np.inf)
    anyz = np.zeros((nth, ntw), bool)
    anynz = np.zeros((nth, ntw), bool)
    # extended index e = r + 1 (e = 0 is the wrap row H-1, which carries depth but no mask cell)
    dext = np.vstack([depth[-1:], depth])
    dext = np.hstack([dext[:, -1:], dext])
    mext = np.zeros((H + 1, W + 1), np.int8) - 1
    mext[1:, 1:] = mask != 0
    for i in range(nth):
        for j in range(ntw):
            blk = dext[i * stride:i * stride + 2 * stride, j * stride:j * stride + 2 * stride]
            mb = mext[i * stride:i * stride + 2 * stride, j * stride:j * stride + 2 * stride]
            zmin[i, j], zmax[i, j] = blk.min(), blk.max()
            anyz[i, j] = (mb == 0).any()
            anynz[i, j] = (mb == 1).any()
    return zmin, zmax, anyz, anynz


def run(seed=0, tile=(2, 32), G=4, SG=16, s1=8, s2=32, H=256, W=256, N=160, t0=0.025, dt=0.005, widen=1, verbose=True):
    depth, mask, albedo, normals, light, amb = bench.synth_faces(1, seed)
    depth, mask = depth[0].astype(np.float64), mask[0]
    _, pt = c_oracle.light_prep(light, clamp_z_min=0.0)
    Cx, Cy, Cz = [float(v) for v in pt[0]]
    rr, cc = np.mgrid[0:H, 0:W]
    x = cc - W / 2.0
    y = H / 2.0 - rr
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
    t = t0 + dt * np.arange(N)
    T1 = tiles(depth, mask, s1)
    T2 = tiles(depth, mask, s2)
    th, tw = tile
    wshape = (H // th, th, W // tw, tw)

    def wave_any(a):
        return a.reshape(wshape).any(axis=(1, 3))

    # per-sample S, masked, rint cell
    S_all = np.empty((N, H, W))
    M_all = np.empty((N, H, W), bool)
    col_all = np.empty((N, H, W), int)
    row_all = np.empty((N, H, W), int)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        col = np.rint(sx).astype(int) + W // 2
        row = H // 2 - np.rint(sy).astype(int)
        col_all[k], row_all[k] = col, row
        colc, rowc = np.clip(col, 0, W - 1), np.clip(row, 0, H - 1)
        M_all[k] = mask[rowc, colc] != 0
        u, v = sx + W / 2.0 - 1e-4, H / 2.0 - sy - 1e-4
        fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
        cu, cv = np.clip(fu + 1, 0, W - 1), np.clip(fv + 1, 0, H - 1)
        wx1, wy1 = u - fu, v - fv
        z = (depth[fv, fu] * (1 - wx1) + depth[fv, cu] * wx1) * (1 - wy1) + \
            (depth[cv, fu] * (1 - wx1) + depth[cv, cu] * wx1) * wy1
        BAx, BAy, BAz = sx - 1e-4 - x, sy + 1e-4 - y, z - zb
        Xx = BAy * BCz - BAz * uy
        Xy = BAz * ux - BAx * BCz
        Xz = BAx * uy - BAy * ux
        S_all[k] = Xx * Xx + Xy * Xy + Xz * Xz

    def lane_skip(ka, kb, T, stride, best):
        """returns (skip, allmasked) per lane for samples ka..kb inclusive"""
        zmin_t, zmax_t, anyz, anynz = T
        cmin = np.minimum(col_all[ka], col_all[kb]) - widen
        cmax = np.maximum(col_all[ka], col_all[kb]) + widen
        rmin = np.minimum(row_all[ka], row_all[kb]) - widen
        rmax = np.maximum(row_all[ka], row_all[kb]) + widen
        tj, ti = np.maximum(cmin, 0) // stride, np.maximum(rmin, 0) // stride
        covered = (cmin >= 0) & (rmin >= 0) & (cmax <= W - 1) & (rmax <= H - 1) & \
                  (cmax + 2 <= (tj + 2) * stride - 1) & (rmax + 2 <= (ti + 2) * stride - 1)
        zmn, zmx = zmin_t[ti, tj], zmax_t[ti, tj]
        allmasked = covered & ~anynz[ti, tj]
        lane_skip.allun = covered & ~anyz[ti, tj]
        Ta, Tb = BCz * t[ka] * proj, BCz * t[kb] * proj
        Tlo, Thi = np.minimum(Ta, Tb), np.maximum(Ta, Tb)
        Pmin, Pmax = n * (zmn - zb), n * (zmx - zb)
        gap = np.maximum(Pmin - Thi, Tlo - Pmax)
        gap0 = np.maximum(-n * zb - Thi, Tlo + n * zb)
        r_ = np.maximum(np.maximum(np.abs(zmn - zb), np.abs(zmx - zb)), np.maximum(np.abs(zb), max(H, W)))
        g = np.minimum(gap, gap0) - (4e-3 * np.abs(BCz) + 1e-6 * np.abs(BCz * proj) * t[-1]
                                     + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * r_)
        cannot = covered & (g > 0) & (g * g * 0.998 > best)
        return allmasked | cannot, allmasked

    best = np.full((H, W), np.inf)
    nw = (H // th) * (W // tw)
    cost_h = 0.0
    cost_1 = 0.0     # single level (current kernel): prefetch every group, test on mask-executed, body if needed
    n_t2 = n_t1 = n_body = 0
    for k0 in range(0, N, SG):
        k1 = min(N, k0 + SG) - 1
        skip2, _ = lane_skip(k0, k1, T2, s2, best)
        need2 = wave_any(~skip2)
        cost_h += nw * C_TEST
        n_t2 += nw
        for g0 in range(k0, k1 + 1, G):
            g1 = min(N, g0 + G) - 1
            skip1, _ = lane_skip(g0, g1, T1, s1, best)
            anyun = M_all[g0:g1 + 1].any(axis=0)
            # hierarchical: level-1 test only in waves whose level-2 test failed
            allun = lane_skip.allun
            need1 = wave_any(~skip1) & need2
            cost_h += need2.sum() * C_TEST
            n_t1 += int(need2.sum())
            # lanes still needing work whose tile is mixed: fetch the exact mask bytes (prefetch-like cost), then decide
            mixed_need = wave_any(~skip1 & ~allun) & need2
            cost_h += mixed_need.sum() * C_PREF
            anyun_ = M_all[g0:g1 + 1].any(axis=0)
            body = wave_any(~skip1 & (allun | anyun_)) & need2
            cost_h += body.sum() * C_BODY
            n_body += int(body.sum())
            n_mixed = locals().get("n_mixed", 0) + int(mixed_need.sum())
            # single level
            ex_mask = wave_any(anyun)
            ex_body = wave_any(anyun & ~skip1)
            cost_1 += nw * C_PREF + ex_mask.sum() * 25 + ex_body.sum() * C_BODY
            # update best with ALL samples (exactness check: skipped lanes must not win)
            Sg = np.where(M_all[g0:g1 + 1], S_all[g0:g1 + 1], np.inf).min(axis=0)
            lane_sk = skip1 | np.repeat(np.repeat(~need2, th, 0), tw, 1)
            lane_sk2 = skip2
            assert not (skip1 & (Sg < best)).any()
            assert not (lane_sk2 & (Sg < best)).any()
            best = np.minimum(best, Sg)
    if verbose:
        print(f"seed {seed} G={G} SG={SG} s1={s1} s2={s2}: tests L2 {n_t2} L1 {n_t1} bodies {n_body} mixed {n_mixed} "
              f"cost hier {cost_h / 1e6:.2f}M  single {cost_1 / 1e6:.2f}M  ratio {cost_h / cost_1:.3f}")
    return cost_h, cost_1


if __name__ == "__main__":
    for (G, SG, s1, s2) in [(4, 16, 8, 32), (4, 8, 8, 16), (4, 16, 8, 16), (2, 8, 8, 16), (4, 32, 8, 64)]:
        a = b = 0
        for s in range(4):
            h, o = run(seed=s, G=G, SG=SG, s1=s1, s2=s2, verbose=False)
            a += h
            b += o
        print(f"G={G} SG={SG} s1={s1} s2={s2}: hier/single = {a / b:.3f}   hier {a / 4e6:.2f}M single {b / 4e6:.2f}M")
