"""Analysis only: wave-level early termination (all lanes either past the mask's bounding box or above the image's
maximum depth and rising).  Counts prefetch groups inside the bbox range with / without termination."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench, c_oracle


def run(seed=0, tile=(8, 8), G=4, H=256, W=256, N=160, t0=0.025, dt=0.005):
    depth, mask, *_rest = bench.synth_faces(1, seed)
    light = _rest[2]
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
    c1 = BCz * proj
    t = t0 + dt * np.arange(N)
    rows, cols = np.nonzero(mask)
    X0, X1 = cols.min() - W / 2.0 - 0.51, cols.max() - W / 2.0 + 0.51
    Y0, Y1 = H / 2.0 - rows.max() - 0.51, H / 2.0 - rows.min() + 0.51
    gz_hi = depth.max()
    err = 4e-3 * np.abs(BCz) + 1e-6 * np.abs(c1) * t[-1] + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * 400
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    wall = lambda a: a.reshape(wsh).all(axis=(1, 3))
    wany = lambda a: a.reshape(wsh).any(axis=(1, 3))
    best = np.full((H, W), np.inf)
    inb_k = np.zeros((N, H, W), bool)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        inb_k[k] = (sx >= X0) & (sx <= X1) & (sy >= Y0) & (sy <= Y1)
    lane_lo = np.where(inb_k.any(0), inb_k.argmax(0), N)
    lane_hi = np.where(inb_k.any(0), N - 1 - inb_k[::-1].argmax(0), -1)
    w_lo = lane_lo.reshape(wsh).min(axis=(1, 3)); w_hi = lane_hi.reshape(wsh).max(axis=(1, 3))
    alive = np.ones(w_lo.shape, bool)
    alive2 = np.ones(w_lo.shape, bool)
    n_pref_rect = 0
    # summed-area style helper: max over rectangles via a sparse table of row-wise/col-wise maxima is overkill here:
    # use a max-pool pyramid with overlapping tiles like the kernel would (stride s, region 2s), levels 8..256
    levels = {}
    for sft in (3, 4, 5, 6, 7, 8):
        st = 1 << sft
        nth, ntw = H // st + 1, W // st + 1
        zm = np.full((nth, ntw), -np.inf)
        for i in range(nth):
            for j in range(ntw):
                blk = depth[i * st:i * st + 2 * st, j * st:j * st + 2 * st]
                if blk.size:
                    zm[i, j] = blk.max()
        levels[sft] = zm
    n_pref = n_pref_early = 0
    for g in range(0, N, G):
        ks = np.arange(g, min(N, g + G))
        in_range = (w_lo <= ks[-1]) & (w_hi >= ks[0])
        n_pref += in_range.sum(); n_pref_early += (in_range & alive).sum(); n_pref_rect += (in_range & alive2).sum()
        Sg = np.full((H, W), np.inf)
        for k in ks:
            sx, sy = x + t[k] * dx, y + t[k] * dy
            col = np.rint(sx).astype(int) + W // 2; row = H // 2 - np.rint(sy).astype(int)
            m = mask[np.clip(row, 0, H - 1), np.clip(col, 0, W - 1)] != 0
            u, v = sx + W / 2.0 - 1e-4, H / 2.0 - sy - 1e-4
            fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
            cu, cv = np.clip(fu + 1, 0, W - 1), np.clip(fv + 1, 0, H - 1)
            wx1, wy1 = u - fu, v - fv
            z = (depth[fv, fu] * (1 - wx1) + depth[fv, cu] * wx1) * (1 - wy1) + (depth[cv, fu] * (1 - wx1) + depth[cv, cu] * wx1) * wy1
            BAx, BAy, BAz = sx - 1e-4 - x, sy + 1e-4 - y, z - zb
            S = (BAy * BCz - BAz * uy) ** 2 + (BAz * ux - BAx * BCz) ** 2 + (BAx * uy - BAy * ux) ** 2
            Sg = np.minimum(Sg, np.where(m, S, np.inf))
        # a wave already terminated must not have had winners
        dead_lane = np.repeat(np.repeat(~alive, th, 0), tw, 1)
        assert not (dead_lane & (Sg < best)).any()
        dead2 = np.repeat(np.repeat(~alive2, th, 0), tw, 1)
        assert not (dead2 & (Sg < best)).any(), int((dead2 & (Sg < best)).sum())
        best = np.minimum(best, Sg)
        knext = ks[-1] + 1
        if knext < N:
            gdone = c1 * t[knext] - n * (gz_hi - zb) - err
            g0 = c1 * t[knext] + n * zb - err            # above the z = 0 plane too
            bound_done = (c1 > 0) & (gdone > 0) & (gdone * gdone * 0.998 > best) & (g0 > 0) & (g0 * g0 * 0.998 > best)
            lane_done = bound_done | (lane_hi < knext)
            alive &= ~wall(lane_done)
            # variant: zmax over the rectangle the rest of the ray can touch (pyramid level that covers it)
            kend = np.clip(lane_hi, knext, N - 1)
            xa, ya = x + t[knext] * dx, y + t[knext] * dy
            xb, yb = x + t[kend] * dx, y + t[kend] * dy
            c0 = np.floor(np.minimum(xa, xb) + W / 2.0).astype(int) - 2; c1_ = np.ceil(np.maximum(xa, xb) + W / 2.0).astype(int) + 2
            r0 = np.floor(H / 2.0 - np.maximum(ya, yb)).astype(int) - 2; r1_ = np.ceil(H / 2.0 - np.minimum(ya, yb)).astype(int) + 2
            c0 = np.clip(c0, 0, W - 1); r0 = np.clip(r0, 0, H - 1); c1_ = np.clip(c1_, 0, W - 1); r1_ = np.clip(r1_, 0, H - 1)
            span = np.maximum(c1_ - c0, r1_ - r0) + 1
            zcap = np.full((H, W), gz_hi)
            for sft in (8, 7, 6, 5, 4, 3):
                st = 1 << sft
                ok = span <= st + 1
                ti, tj = r0 // st, c0 // st
                zc = levels[sft][np.minimum(ti, levels[sft].shape[0] - 1), np.minimum(tj, levels[sft].shape[1] - 1)]
                zcap = np.where(ok, zc, zcap)
            zcap = np.maximum(zcap, 0.0)
            gdone2 = c1 * t[knext] - n * (zcap - zb) - err
            bound_done2 = (c1 > 0) & (gdone2 > 0) & (gdone2 * gdone2 * 0.998 > best)
            alive2 &= ~wall(bound_done2 | (lane_hi < knext))
    print(f"seed {seed}: prefetch groups in bbox range {n_pref}  with early termination {n_pref_early}  ({n_pref_early / n_pref:.3f})  remaining-rectangle zmax {n_pref_rect} ({n_pref_rect / n_pref:.3f})")


for s in range(4):
    run(seed=s)
