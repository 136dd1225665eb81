"""Analysis only (CPU, numpy f64 -- not a parity tool): what a cap that knows where the ray ENDS would add to the horizon tables.

tools/sim_horizon.py prices the library's termination cap (running maxima of the readable depth from the ray's current column /
row to the image border).  A ray stops at t = 0.82 of the way to its end point, so the columns / rows it can still touch form a
RANGE [current cell, last sample's cell]; this model counts visited sample groups (16 x 4 wave tiles) with the tables and with
1-D range maxima over exactly those columns and rows (a sparse table per axis would serve them: two look-ups per axis).
Result, 3 synthetic + 3 FFHQ-fixture faces: 58 938 -> 49 434 visits (-16 %); the full remaining RECTANGLE (2-D range maximum)
gives -20.5 %, a quadrant maximum (2-D running maximum towards the light) -0.6 %.  Priced in DESIGN.md section 7, not built.
"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
from scipy.ndimage import binary_dilation
import bench, c_oracle

def build_st(a):
    H, W = a.shape
    LH, LW = int(np.log2(H)) + 1, int(np.log2(W)) + 1
    st = {}
    st[(0, 0)] = a
    for i in range(LH):
        if i > 0:
            p = st[(i - 1, 0)]; sh = 1 << (i - 1)
            q = p.copy(); q[:H - sh] = np.maximum(p[:H - sh], p[sh:]); st[(i, 0)] = q
        for j in range(1, LW):
            p = st[(i, j - 1)]; sh = 1 << (j - 1)
            q = p.copy(); q[:, :W - sh] = np.maximum(p[:, :W - sh], p[:, sh:]); st[(i, j)] = q
    return st

def qmax(st, r0, r1, c0, c1):
    # inclusive ranges, arrays
    kh = np.floor(np.log2(r1 - r0 + 1)).astype(int); kw = np.floor(np.log2(c1 - c0 + 1)).astype(int)
    out = np.zeros(r0.shape)
    for i in np.unique(kh):
        for j in np.unique(kw):
            m = (kh == i) & (kw == j)
            if not m.any(): continue
            t = st[(i, j)]
            a, b = r0[m], c0[m]; a2, b2 = r1[m] - (1 << i) + 1, c1[m] - (1 << j) + 1
            out[m] = np.maximum(np.maximum(t[a, b], t[a, b2]), np.maximum(t[a2, b], t[a2, b2]))
    return out

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
    err = 4e-3 * np.abs(BCz) + 1e-6 * np.abs(c1) * t[-1] + (1e-6 * n + 2e-7 * (np.abs(ux) + np.abs(uy) + np.abs(BCz))) * 400
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    wall = lambda a: a.reshape(wsh).all(axis=(1, 3))
    best = np.full((H, W), np.inf)
    inb_k = np.zeros((N, H, W), bool)
    for k in range(N):
        sx, sy = x + t[k] * dx, y + t[k] * dy
        inb_k[k] = (sx >= X0) & (sx <= X1) & (sy >= Y0) & (sy <= Y1)
    lane_lo = np.where(inb_k.any(0), inb_k.argmax(0), N)
    lane_hi = np.where(inb_k.any(0), N - 1 - inb_k[::-1].argmax(0), -1)
    w_lo = lane_lo.reshape(wsh).min(axis=(1, 3)); w_hi = lane_hi.reshape(wsh).max(axis=(1, 3))
    alive3 = np.ones(w_lo.shape, bool); alive4 = alive3.copy()
    md = binary_dilation(mask != 0, structure=np.ones((3, 3)), iterations=2)
    dpad = np.where(md, np.maximum(depth, 0.0), 0.0)
    gz = dpad.max()
    colmax = dpad.max(axis=0); rowmax = dpad.max(axis=1)
    col_suf = np.maximum.accumulate(colmax[::-1])[::-1]; col_pre = np.maximum.accumulate(colmax)
    row_suf = np.maximum.accumulate(rowmax[::-1])[::-1]; row_pre = np.maximum.accumulate(rowmax)
    stc = build_st(np.tile(colmax[None, :], (2, 1))); str_ = build_st(np.tile(rowmax[:, None], (1, 2)))
    xe, ye = x + t[-1] * dx, y + t[-1] * dy
    n3 = n4 = 0
    for g in range(0, N, G):
        ks = np.arange(g, min(N, g + G))
        in_range = (w_lo <= ks[-1]) & (w_hi >= ks[0])
        n3 += (in_range & alive3).sum(); n4 += (in_range & alive4).sum()
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
        for al in (alive3, alive4):
            dead = np.repeat(np.repeat(~al, th, 0), tw, 1)
            assert not (dead & (Sg < best)).any()
        best = np.minimum(best, Sg)
        knext = ks[-1] + 1
        if knext < N and (g // G) % 2 == 1:
            xa, ya = x + t[knext] * dx, y + t[knext] * dy
            ca = np.clip(np.floor(xa + W / 2.0).astype(int), 0, W - 1); ra = np.clip(np.floor(H / 2.0 - ya).astype(int), 0, H - 1)
            zc = np.where(dx >= 0, col_suf[np.maximum(ca - 2, 0)], col_pre[np.minimum(ca + 3, W - 1)])
            zr = np.where(dy <= 0, row_suf[np.maximum(ra - 2, 0)], row_pre[np.minimum(ra + 3, H - 1)])
            zcap = np.minimum(zc, zr)
            gd3 = c1 * t[knext] - n * (zcap - zb) - err
            alive3 &= ~wall(((c1 > 0) & (gd3 > 0) & (gd3 * gd3 * 0.998 > best)) | (lane_hi < knext))
            ce = np.clip(np.floor(xe + W / 2.0).astype(int), 0, W - 1); re_ = np.clip(np.floor(H / 2.0 - ye).astype(int), 0, H - 1)
            c0 = np.clip(np.minimum(ca, ce) - 2, 0, W - 1); c1_ = np.clip(np.maximum(ca, ce) + 3, 0, W - 1)
            r0 = np.clip(np.minimum(ra, re_) - 2, 0, H - 1); r1_ = np.clip(np.maximum(ra, re_) + 3, 0, H - 1)
            z2 = np.minimum(qmax(stc, np.zeros_like(c0), np.zeros_like(c0), c0, c1_), qmax(str_, r0, r1_, np.zeros_like(r0), np.zeros_like(r0)))
            gd4 = c1 * t[knext] - n * (z2 - zb) - err
            alive4 &= ~wall(((c1 > 0) & (gd4 > 0) & (gd4 * gd4 * 0.998 > best)) | (lane_hi < knext))
    print(f"{data} seed {seed}: row/col tables {n3}, 1-D range maxima (columns, rows) {n4} ({n4 / n3:.3f})")
    return n3, n4

tot = np.zeros(2)
for d in ("synthetic", "ffhq"):
    for s in range(3):
        tot += run(seed=s, data=d)
print(tot, tot[1] / tot[0])
