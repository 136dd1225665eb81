import sys, os, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from geomconsistentfr_amd import RenderParams, _lib, light_prep, shadow_min_distance
dev = torch.device("cuda:0")
for (Hs, Ws, N, dt) in [(96, 64, 33, 0.024), (128, 128, 80, 0.01), (64, 96, 40, 0.02)]:
    rng = np.random.default_rng(Hs * 7 + Ws)
    r, c = np.mgrid[0:Hs, 0:Ws]
    bump = 0.35 * Hs * np.exp(-(((c - 0.5 * Ws) / (0.25 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.3 * Hs)) ** 2))
    depth = np.stack([bump, bump + 3 * rng.random((Hs, Ws)), 30 * rng.random((Hs, Ws)), bump - 0.15 * Hs, -bump, bump + 1000.0]).astype(np.float32)
    B = depth.shape[0]
    ell = ((((c - 0.5 * Ws) / (0.4 * Ws)) ** 2 + ((r - 0.5 * Hs) / (0.45 * Hs)) ** 2) < 1)
    mask = np.stack([ell, ell, np.ones_like(ell), rng.random((Hs, Ws)) > 0.2, ell, rng.random((Hs, Ws)) > 0.7]).astype(np.uint8)
    lights = np.array([[[0.75, 0.0, 0.66], [0.1, -0.2, 0.97]]] * B, np.float32)
    lights[1::2, 0] = [-0.5, 0.47, 0.72]
    lights[2, 1] = [0.99, 0.05, 0.05]
    prm = RenderParams(n_samples=N, t0=0.025, dt=dt)
    t = lambda a: torch.from_numpy(a).to(dev)
    _, pt = light_prep(t(lights), prm)
    for zb in (1, 0):
        for want in (True, False):
            for rep in range(2):
                a, _ = shadow_min_distance(t(depth), t(mask), pt, prm, want_argmin=want, options=_lib.options(lds_stage=0, ksplit=0, depth_bound_skip=zb))
                b, _ = shadow_min_distance(t(depth), t(mask), pt, prm, want_argmin=want, options=_lib.options(lds_stage=1, ksplit=0, depth_bound_skip=zb))
                d = (a != b) & ~(torch.isnan(a) & torch.isnan(b))
                per = d.reshape(B, 2, -1).sum(-1).cpu().numpy()
                rows = d.any(-1).any(1).cpu().numpy()          # (B, H)
                print((Hs, Ws, N), "zb", zb, "argmin", want, "rep", rep, "diff per (image,light):", per.tolist(),
                      "rows with diffs per image:", [np.nonzero(rows[i])[0][[0, -1]].tolist() if rows[i].any() else None for i in range(B)])
