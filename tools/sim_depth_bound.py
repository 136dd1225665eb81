"""Analysis only: how many wave-groups of the march would a conservative depth-bound ("hierarchical z") test
remove, on top of the mask-group skip the kernel already has?  Numpy, f64, statistics -- not a parity tool.

For pixel B, light C, sample A_k = (s_k, z_bilinear):  S_k = |BA x BC|^2 >= n^2 (BAz - BCz * alpha_k)^2 with
n = |BC_xy|, alpha_k = (BA_xy . BC_xy)/n^2.  With z in [zmin, zmax] over the group's footprint the bound is
evaluated at the group's first/last sample; a lane votes "skip" when the bound exceeds its running minimum.
"""
import sys

import numpy as np

sys.path.insert(0, "/root/repo")
sys.path.insert(0, "/root/repo/oracle")
import bench  # noqa: E402
import c_oracle  # noqa: E402


def run(seed=0, tile=(2, 32), depth_group=4, stride=8, H=256, W=256, N=160, t0=0.025, dt=0.005):
    depth, mask, albedo, normals, light, amb = bench.synth_faces(1, seed)
    depth, mask = depth[0].astype(np.float64), mask[0]
    _, pt = c_oracle.light_prep(light, clamp_z_min=0.0)
    Cx, Cy, Cz = [float(v) for v in pt[0]]
    rr, cc = np.mgrid[0:H, 0:W]
    x = cc - W / 2.0
    y = H / 2.0 - rr
    # end point: intersection of the ray towards (Cx, Cy) with the box [-W/2, W/2-1] x [-H/2+1, H/2]
    ux, uy = Cx - x, Cy - y
    with np.errstate(divide="ignore", invalid="ignore"):
        tx = np.where(ux > 0, (W / 2.0 - 1 - x) / ux, np.where(ux < 0, (-W / 2.0 - x) / ux, np.inf))
        ty = np.where(uy > 0, (H / 2.0 - y) / uy, np.where(uy < 0, (-H / 2.0 + 1 - y) / uy, np.inf))
    te = np.minimum(np.minimum(tx, ty), 1.0)
    dx, dy = te * ux, te * uy
    zb = depth
    BCz = Cz - zb
    n = np.sqrt(ux * ux + uy * uy)
    t = t0 + dt * np.arange(N)

    # overlapped min/max tiles: tile (i, j) covers [stride*i, stride*i + 2*stride)
    nth, ntw = H // stride, W // stride
    pad = np.pad(depth, ((0, 2 * stride), (0, 2 * stride)), mode="edge")
    zmin_t = np.empty((nth, ntw))
    zmax_t = np.empty((nth, ntw))
    for i in range(nth):
        for j in range(ntw):
            blk = pad[i * stride:i * stride + 2 * stride, j * stride:j * stride + 2 * stride]
            zmin_t[i, j], zmax_t[i, j] = blk.min(), blk.max()

    best = np.full((H, W), np.inf)
    th, tw = tile
    n_groups = (N + depth_group - 1) // depth_group
    exec_now = np.zeros((H // th, W // tw), np.int64)
    exec_new = np.zeros((H // th, W // tw), np.int64)
    exec_perf = np.zeros((H // th, W // tw), np.int64)
    uncovered = 0
    for g in range(n_groups):
        ks = np.arange(g * depth_group, min(N, (g + 1) * depth_group))
        any_unmasked = np.zeros((H, W), bool)
        S_g = []
        cols, rows = [], []
        for k in ks:
            sx, sy = x + t[k] * dx, y + t[k] * dy
            col = np.rint(sx).astype(int) + W // 2
            row = H // 2 - np.rint(sy).astype(int)
            inb = (col >= 0) & (col < W) & (row >= 0) & (row < H)
            m = np.zeros((H, W), bool)
            m[inb] = mask[row[inb], col[inb]] != 0
            u, v = sx + W / 2.0 - 1e-4, H / 2.0 - sy - 1e-4
            fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
            cols.append(fu)
            rows.append(fv)
            cu, cv = np.clip(fu + 1, 0, W - 1), np.clip(fv + 1, 0, H - 1)
            fu_, fv_ = np.clip(fu, 0, W - 1), np.clip(fv, 0, H - 1)
            wx1, wy1 = u - fu, v - fv
            z = (depth[fv_, fu_] * (1 - wx1) + depth[fv_, cu] * wx1) * (1 - wy1) + \
                (depth[cv, fu_] * (1 - wx1) + depth[cv, cu] * wx1) * wy1
            BAx, BAy, BAz = sx - 1e-4 - x, sy + 1e-4 - y, z - zb
            Xx = BAy * BCz - BAz * uy
            Xy = BAz * ux - BAx * BCz
            Xz = BAx * uy - BAy * ux
            S = Xx * Xx + Xy * Xy + Xz * Xz
            S_g.append(np.where(m, S, np.inf))
            any_unmasked |= m
        # depth bound of the group, per lane
        c0 = np.minimum(cols[0], cols[-1])
        c1 = np.maximum(cols[0], cols[-1]) + 1
        r0 = np.minimum(rows[0], rows[-1])
        r1 = np.maximum(rows[0], rows[-1]) + 1
        ti = np.clip(r0, 0, H - 1) // stride
        tj = np.clip(c0, 0, W - 1) // stride
        ti, tj = np.minimum(ti, nth - 1), np.minimum(tj, ntw - 1)
        covered = (r1 <= ti * stride + 2 * stride - 1) & (c1 <= tj * stride + 2 * stride - 1) & (r0 >= 0) & (c0 >= 0)
        zmn, zmx = zmin_t[ti, tj], zmax_t[ti, tj]
        # G(k, z) = n (z - zb) - BCz * alpha_k * n,   alpha_k n = t_k (d . u)/n
        proj = (dx * ux + dy * uy) / np.maximum(n, 1e-9)
        Gs = []
        for k in (ks[0], ks[-1]):
            for z in (zmn, zmx):
                Gs.append(n * (z - zb) - BCz * t[k] * proj)
        Gs = np.stack(Gs)
        same_sign = np.all(Gs > 0, axis=0) | np.all(Gs < 0, axis=0)
        gmin = np.abs(Gs).min(axis=0)
        err = 4e-3 * np.abs(BCz) + 1e-6 * (np.abs(BCz) + n) * (t[ks[-1]] * np.hypot(dx, dy) + 100.0)
        can_skip = covered & same_sign & ((gmin - err) > 1.001 * np.sqrt(best)) & np.isfinite(best)
        uncovered += int((~covered & any_unmasked).sum())
        need_now = any_unmasked
        need_new = any_unmasked & ~can_skip
        exec_now += need_now.reshape(H // th, th, W // tw, tw).any(axis=(1, 3))
        exec_new += need_new.reshape(H // th, th, W // tw, tw).any(axis=(1, 3))
        # sanity: a skipped lane's samples must not beat its running minimum
        Smin = np.minimum.reduce(S_g)
        bad = can_skip & (Smin < best)
        assert not bad.any(), (g, int(bad.sum()))
        exec_perf += (Smin < best).reshape(H // th, th, W // tw, tw).any(axis=(1, 3))
        best = np.minimum(best, Smin)
    run.last = dict(now=exec_now.copy(), new=exec_new.copy(), perf=exec_perf.copy())
    total = n_groups * exec_now.size
    print(f"seed {seed} light {light[0]}  wave-groups: all {total}  mask-skip {exec_now.sum()} "
          f"({exec_now.sum() / total:.3f})  +depth-bound {exec_new.sum()} ({exec_new.sum() / total:.3f})  "
          f"perfect {exec_perf.sum() / total:.3f}  ratio {exec_new.sum() / max(1, exec_now.sum()):.3f}  uncovered lanes {uncovered}")
    return exec_now.sum(), exec_new.sum()


if __name__ == "__main__":
    a = b = 0
    for s in range(8):
        n0, n1 = run(seed=s)
        a += n0
        b += n1
    print("overall ratio", b / a)

