"""Analysis only (CPU, numpy f64 -- not a parity tool): what pruning a ray's sample range against the mask's bounding OCTAGON
(the box plus the extents of column + row and column - row over the non-zero cells) saves over the box alone.

Counts, per face and for 16 x 4 wave tiles, the sample groups a wave visits with the horizon-table termination in place
(tools/sim_horizon.py) when the per-lane candidate range [lane_lo, lane_hi] comes from the box and when it comes from the
octagon.  Result, 3 synthetic + 3 FFHQ-fixture faces: 59 430 -> 48 978 visited groups (-17.6 %; an ellipse's octagon cuts four
fifths of its box's corners).  Built in round 3 (march prologue + the prepass' statistics job).
"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench, c_oracle

def run(seed=0, tile=(4, 16), G=4, H=256, W=256, N=160, t0=0.025, dt=0.005, data="synthetic"):
    if data == "ffhq":
        depth, mask, _a, _n, light, _amb = bench.ffhq_faces(1, seed)
    else:
        depth, mask, _a, _n, light, _amb = bench.synth_faces(1, seed)
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
    gz_hi = max(depth.max(), 0.0)
    err = 4e-3 * np.abs(BCz) + 1e-6 * np.abs(c1) * t[-1] + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * 400
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    wall = lambda a: a.reshape(wsh).all(axis=(1, 3))
    best = np.full((H, W), np.inf)
    inb_k = np.zeros((N, H, W), bool)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        inb_k[k] = (sx >= X0) & (sx <= X1) & (sy >= Y0) & (sy <= Y1)
    S0, S1 = (cols + rows).min() - 1.02, (cols + rows).max() + 1.02
    D0, D1 = (cols - rows).min() - 1.02, (cols - rows).max() + 1.02
    inb_o = np.zeros((N, H, W), bool)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        cs, rs = sx + W / 2.0, H / 2.0 - sy
        inb_o[k] = inb_k[k] & (cs + rs >= S0) & (cs + rs <= S1) & (cs - rs >= D0) & (cs - rs <= D1)
    lane_lo_o = np.where(inb_o.any(0), inb_o.argmax(0), N)
    lane_hi_o = np.where(inb_o.any(0), N - 1 - inb_o[::-1].argmax(0), -1)
    w_lo_o = lane_lo_o.reshape(wsh).min(axis=(1, 3)); w_hi_o = lane_hi_o.reshape(wsh).max(axis=(1, 3))
    alive5 = np.ones(w_lo_o.shape, bool); n5 = 0
    lane_lo = np.where(inb_k.any(0), inb_k.argmax(0), N)
    lane_hi = np.where(inb_k.any(0), N - 1 - inb_k[::-1].argmax(0), -1)
    w_lo = lane_lo.reshape(wsh).min(axis=(1, 3)); w_hi = lane_hi.reshape(wsh).max(axis=(1, 3))
    alive = np.ones(w_lo.shape, bool); alive2 = alive.copy(); alive3 = alive.copy()
    # column / row maxima with a 2-cell dilation (bilinear footprint), then running maxima from each side
    from scipy.ndimage import binary_dilation
    md = binary_dilation(mask != 0, structure=np.ones((3, 3)), iterations=2)
    dpad = np.where(md, np.maximum(depth, 0.0), 0.0)
    gz_hi_m = dpad.max()
    colmax = dpad.max(axis=0); rowmax = dpad.max(axis=1)
    dil = lambda v: np.maximum.reduce([np.roll(v, s) for s in (-2, -1, 0, 1, 2)])
    colmax, rowmax = dil(colmax), dil(rowmax)   # (roll wraps: conservative enough for a model)
    col_suf = np.maximum.accumulate(colmax[::-1])[::-1]; col_pre = np.maximum.accumulate(colmax)
    row_suf = np.maximum.accumulate(rowmax[::-1])[::-1]; row_pre = np.maximum.accumulate(rowmax)
    n1 = n2 = n3 = 0
    for g in range(0, N, G):
        ks = np.arange(g, min(N, g + G))
        in_range = (w_lo <= ks[-1]) & (w_hi >= ks[0])
        n1 += (in_range & alive).sum(); n3 += (in_range & alive3).sum()
        in_range_o = (w_lo_o <= ks[-1]) & (w_hi_o >= ks[0]); n5 += (in_range_o & alive5).sum()
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
        dead3 = np.repeat(np.repeat(~alive3, th, 0), tw, 1)
        assert not (dead3 & (Sg < best)).any(), int((dead3 & (Sg < best)).sum())
        best = np.minimum(best, Sg)
        knext = ks[-1] + 1
        if knext < N and (g // G) % 2 == 1:
            gdone = c1 * t[knext] - n * (gz_hi - zb) - err
            done = ((c1 > 0) & (gdone > 0) & (gdone * gdone * 0.998 > best)) | (lane_hi < knext)
            alive &= ~wall(done)
            # row / column running maxima from the next sample's cell towards where the ray goes
            xa, ya = x + t[knext] * dx, y + t[knext] * dy
            ca = np.clip(np.floor(xa + W / 2.0).astype(int), 0, W - 1); ra = np.clip(np.floor(H / 2.0 - ya).astype(int), 0, H - 1)
            zc = np.where(dx >= 0, col_suf[np.maximum(ca - 1, 0)], col_pre[np.minimum(ca + 2, W - 1)])
            zr = np.where(dy <= 0, row_suf[np.maximum(ra - 1, 0)], row_pre[np.minimum(ra + 2, H - 1)])
            zcap = np.minimum(np.minimum(zc, zr), gz_hi_m)
            gd3 = c1 * t[knext] - n * (zcap - zb) - err
            done3 = ((c1 > 0) & (gd3 > 0) & (gd3 * gd3 * 0.998 > best)) | (lane_hi < knext)
            alive3 &= ~wall(done3)
            alive5 &= ~wall(((c1 > 0) & (gd3 > 0) & (gd3 * gd3 * 0.998 > best)) | (lane_hi_o < knext))
    print(f"{data} seed {seed}: tables {n3}, tables + octagon range {n5} ({n5 / n3:.3f})")
    return n3, n5

tot = np.zeros(2)
for d in ("synthetic", "ffhq"):
    for s in range(3):
        tot += run(seed=s, data=d)
print(tot, tot[1] / tot[0])
