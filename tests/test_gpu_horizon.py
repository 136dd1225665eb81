"""GPU: the prepass' horizon tables (csrc/gcfr_shadow.hip build_horizon_block) -- the running column / row maxima the trailing
loop of the march terminates against -- read back from the workspace and compared with their definition in numpy.

The tables must BOUND, from above, every depth value an unmasked sample can read (cells within one cell of a non-zero mask
cell, T8:472-494) and 0; the kernel dilates the mask by whole dwords of columns (a superset).  Checked here: equality with
the numpy statement of exactly that construction (the kernel stores, per entry, the values of four row bands whose maximum
is the table), and >= the tight 3 x 3 dilation it is a superset of.  That the march's
results do not change is tests/test_gpu_parity.py's business (bit-equality to the C oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
S = 1024  # kHorizonDim


def _tables(depth, mask):
    """(col_pre, col_suf, row_pre, row_suf) as the kernel defines them, plus the tight column / row maxima."""
    H, W = depth.shape
    nz = mask != 0
    z = np.where(np.isnan(depth), 0.0, np.maximum(depth, 0.0)).astype(np.float32)
    # kernel: per row one flag per dword of four columns, dilated by one dword and one row; with more than one 256-column
    # segment the segment's edge dwords are always live
    dw = nz.reshape(H, W // 4, 4).any(axis=2)
    v3 = dw.copy()
    v3[1:] |= dw[:-1]
    v3[:-1] |= dw[1:]
    live = v3.copy()
    nseg = (W + 255) // 256
    for s in range(nseg):
        seg = v3[:, s * 64:(s + 1) * 64]
        out = seg.copy()
        out[:, 1:] |= seg[:, :-1]
        out[:, :-1] |= seg[:, 1:]
        if nseg > 1:
            out[:, 0] = True
            if out.shape[1] == 64:
                out[:, 63] = True
        live[:, s * 64:(s + 1) * 64] = out
    live = np.repeat(live, 4, axis=1)
    tight = nz.copy()
    for dr in (-1, 0, 1):
        for dc in (-1, 0, 1):
            sh = np.zeros_like(nz)
            sh[max(dr, 0):H + min(dr, 0), max(dc, 0):W + min(dc, 0)] = nz[max(-dr, 0):H + min(-dr, 0), max(-dc, 0):W + min(-dc, 0)]
            tight |= sh
    assert (live | ~tight).all()                      # the kernel's live set contains the tight one

    def scans(v):
        v = v.copy()
        v[0] = v[-1] = max(v[0], v[-1])               # the corners' wrap partners, folded into both ends
        n = v.size
        pre, suf = np.maximum.accumulate(v), np.maximum.accumulate(v[::-1])[::-1]
        j = np.clip(np.arange(S) - S // 2 + n // 2, 0, n - 1)
        return pre[j], suf[j]
    zl, zt = np.where(live, z, 0.0), np.where(tight, z, 0.0)
    cp, cs = scans(zl.max(axis=0))
    rp, rs = scans(zl.max(axis=1))
    tcp, tcs = scans(zt.max(axis=0))
    trp, trs = scans(zt.max(axis=1))
    return (cp, cs, rp, rs), (tcp, tcs, trp, trs)


@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (5, 64, 96), (3, 130, 100), (1, 512, 512)])
def test_horizon_tables_match_their_definition(B, H, W):
    from geomconsistentfr_amd import _lib, block as R, RenderParams
    L_ = _lib.load()
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(B * 1000 + H)
    depth = (rng.standard_normal((B, H, W)) * 40.0 + 10.0).astype(np.float32)
    depth[:, H // 3, W // 5] = np.nan
    depth[0, 0, 0] = 500.0                            # a peak in a corner cell: reaches the tables only where the mask says so
    mask = np.zeros((B, H, W), np.uint8)
    for b in range(B):
        r0, c0 = rng.integers(0, H // 2), rng.integers(0, W // 2)
        mask[b, r0:r0 + H // 3, c0:c0 + W // 3] = (rng.random((H // 3, W // 3)) < 0.6) * 255
    mask[0, 0, 1] = 1                                 # ... image 0: the corner is live (and wraps)
    mask[B - 1, H - 1, W - 1] = 7
    d, m = torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev)
    light = torch.tensor([[[1200.0, 2400.0, 3000.0]]], device=dev).repeat(B, 1, 1)
    prm = RenderParams()
    tt = R.sample_table(prm, dev)
    md = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    ws_bytes = int(L_.gcfr_shadow_workspace_bytes(B, H, W))
    ws = torch.zeros(ws_bytes, dtype=torch.uint8, device=dev)
    opt = _lib.options(ksplit=0)                      # the grid schedule (tiny launches would pick the k-split: no trailing loop, no tables)
    _lib.check(L_.gcfr_shadow_fwd(d.data_ptr(), m.data_ptr(), B, light.data_ptr(), B, 1, H, W, prm.n_samples, tt.data_ptr(), 0.0, None,
                                  md.data_ptr(), None, ws.data_ptr(), ws_bytes, None, _lib.opt_ref(opt)), "gcfr_shadow_fwd")
    torch.cuda.synchronize()
    n_stat = (H * W + 16383) // 16384
    zb_stride = ((((H >> 3) + 1) * ((W >> 3) + 1) + 1) + 63) & ~63
    base = B * (H + 1) * (W + 1) * 16 + B * n_stat * 16
    raw = ws.cpu().numpy()
    for b in range(B):
        off = base + (b * (zb_stride + 4 * S) + zb_stride) * 16
        got = raw[off:off + 4 * S * 16].view(np.float32).reshape(4, S, 4).max(axis=2)   # an entry: the four row bands' values
        want, tight = _tables(depth[b], mask[b])
        for k, name in enumerate(("col_pre", "col_suf", "row_pre", "row_suf")):
            assert np.array_equal(got[k], want[k]), (b, name, np.flatnonzero(got[k] != want[k])[:8])
            assert (got[k] >= tight[k]).all(), (b, name)
    assert float(got[1][0]) == float(want[1].max())   # col_suf's first entry: the cap of the main loop's termination test


@pytest.mark.parametrize("H,W", [(2, 4), (4, 8), (6, 12), (10, 20), (34, 36), (18, 260)])
def test_tiny_and_odd_shapes_march_bit_identically_with_the_tables(H, W):
    """Shapes whose row bands are empty or one row high, a width with a partial and a second 256-column segment: the grid
    schedule (trailing loop + horizon tables, forced -- launches this small would pick the k-split) against the C oracle."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import c_oracle
    from geomconsistentfr_amd import _lib, RenderParams, light_prep, shadow_min_distance
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(H * 1000 + W)
    B, N = 6, 40
    r, c = np.mgrid[0:H, 0:W]
    depth = (0.4 * H * np.exp(-(((c - 0.5 * W) / (0.3 * W)) ** 2 + ((r - 0.5 * H) / (0.3 * H)) ** 2)))[None].repeat(B, 0)
    depth = (depth + rng.random((B, H, W)) * 0.5).astype(np.float32)
    depth[1] = -depth[1]
    depth[2, 0, :] += 50.0                                  # a wall along the top row (wrap partner of the bottom one)
    mask = (rng.random((B, H, W)) < 0.7).astype(np.uint8)
    mask[3] = 1
    mask[4, :, : W // 2] = 0
    lights = rng.standard_normal((B, 3)).astype(np.float32)
    lights[5] = (0.0, 0.3, 0.9)
    prm = RenderParams(n_samples=N, dt=0.8 / N)
    _, pt = light_prep(torch.from_numpy(lights).to(dev), prm)
    md, am = shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev), pt.reshape(B, 1, 3), prm,
                                 options=_lib.options(ksplit=0))
    _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(0.025, 0.8 / N, N))
    md, am = md.cpu().numpy(), am.cpu().numpy()
    assert np.array_equal(md, md_o)
    lit = md_o < 1e5
    assert np.array_equal(am[lit], am_o[lit])
