"""Analysis only: how many prefetch stages could skip the mask-byte gathers if tiles carried a "deep interior"
flag (every mask cell within the tile region grown by one stride is set)?"""
import sys
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/oracle")
import bench, c_oracle


def run(seed=0, tile=(8, 8), G=4, s=8, grow=8, H=256, W=256, N=160, t0=0.025, dt=0.005):
    depth, mask, *_rest = bench.synth_faces(1, seed)
    light = _rest[2]
    mask = mask[0] != 0
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
    t = t0 + dt * np.arange(N)
    nth, ntw = H // s + 1, W // s + 1
    mext = np.zeros((H + 1 + 4 * s, W + 1 + 4 * s), bool)          # extended grid, padded with "masked"
    mext[grow + 1:grow + 1 + H, grow + 1:grow + 1 + W] = mask
    D = np.zeros((nth, ntw), bool)
    for i in range(nth):
        for j in range(ntw):
            blk = mext[i * s:i * s + 2 * s + 2 * grow, j * s:j * s + 2 * s + 2 * grow]   # region grown by `grow`
            D[i, j] = blk.all()
    rows, cols = np.nonzero(mask)
    X0, X1 = cols.min() - W / 2.0 - 0.51, cols.max() - W / 2.0 + 0.51
    Y0, Y1 = H / 2.0 - rows.max() - 0.51, H / 2.0 - rows.min() + 0.51
    th, tw = tile
    wsh = (H // th, th, W // tw, tw)
    n_in = n_D = 0
    for g in range(0, N - G, G):
        ka, kb = g, g + G - 1
        sxa, sya = x + t[ka] * dx, y + t[ka] * dy
        sxb, syb = x + t[kb] * dx, y + t[kb] * dy
        inb = ((sxa >= X0) & (sxa <= X1) & (sya >= Y0) & (sya <= Y1)) | ((sxb >= X0) & (sxb <= X1) & (syb >= Y0) & (syb <= Y1))
        ca = np.rint(sxa).astype(int) + W // 2; cb = np.rint(sxb).astype(int) + W // 2
        ra = H // 2 - np.rint(sya).astype(int); rb = H // 2 - np.rint(syb).astype(int)
        tj = np.clip(np.minimum(ca, cb), 0, W - 1) // s; ti = np.clip(np.minimum(ra, rb), 0, H - 1) // s
        d = D[np.minimum(ti, nth - 1), np.minimum(tj, ntw - 1)]
        w_in = inb.reshape(wsh).any(axis=(1, 3))
        w_D = d.reshape(wsh).all(axis=(1, 3))
        n_in += w_in.sum(); n_D += (w_in & w_D).sum()
    print(f"seed {seed}: wave-groups in mask-box range {n_in}, all lanes deep-interior {n_D} ({n_D / n_in:.3f})")


for sd in range(3):
    run(seed=sd)
    run(seed=sd, grow=4)
