"""GPU: the prepass' horizon tables (csrc/gcfr_shadow.hip build_horizon_block) -- the running column / row maxima the trailing
loop of the march terminates against -- read back from the workspace and compared with their definition in numpy.

The tables must BOUND, from above, every depth value an unmasked sample can read (cells within one cell of a non-zero mask
cell, T8:472-494) and 0; the kernel dilates the mask by whole dwords of columns (a superset).  Checked here: equality with
the numpy statement of exactly that construction (the kernel stores, per entry, the values of four row bands whose maximum
is the table), and >= the tight 3 x 3 dilation (wrapping at index -1 as the reference's gathers do) it is a superset of.  That the march's
results do not change is tests/test_gpu_parity.py's business (bit-equality to the C oracle)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
S = 1024  # kHorizonDim
BACK, AHEAD = 2, 3   # cells of slack built into the suffix / prefix entries (gcfr_shadow.hip build_horizon_block)


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
    v3[H - 1] |= dw[0]                                # the dilation wraps where the gathers do (index -1 == last, T8:488-491):
    live = v3.copy()                                  # row 0 unmasked -> row H-1 is readable
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
    if nseg > 1:
        live[:, -1] = True                            # ... the image's last dword of columns: always live with several segments,
    else:
        live[:, -1] |= v3[:, 0]                       # the wrap partner of column 0 otherwise
    live = np.repeat(live, 4, axis=1)
    # the cells an unmasked sample can read: its rounded cell's 3 x 3 neighbourhood, where index -1 is the LAST column / row
    # (floor(u) = -1 wraps, T8:488-491) and index W / H does not occur (the end point is clamped to the box, T8:462-465)
    tight = np.zeros_like(nz)
    for dr in (-1, 0, 1):
        for dc in (-1, 0, 1):
            sh = np.roll(np.roll(nz, dr, axis=0), dc, axis=1)
            if dr == 1:
                sh[0] = False
            if dc == 1:
                sh[:, 0] = False
            tight |= sh
    assert (live | ~tight).all()                      # the kernel's live set contains the tight one

    def scans(v, back=BACK, ahead=AHEAD):
        v = v.copy()
        v[0] = v[-1] = max(v[0], v[-1])               # the corners' wrap partners, folded into both ends
        n = v.size
        pre, suf = np.maximum.accumulate(v), np.maximum.accumulate(v[::-1])[::-1]
        c0 = np.arange(S) - S // 2 + n // 2           # the cell entry i belongs to (outside the image: the nearest one)
        # the look-up's slack is part of the table (round 5): the prefix entry of cell c covers the cells <= c + 3, the suffix
        # entry the cells >= c - 2
        return pre[np.clip(c0 + ahead, 0, n - 1)], suf[np.clip(c0 - back, 0, n - 1)]
    zl, zt = np.where(live, z, 0.0), np.where(tight, z, 0.0)
    cp, cs = scans(zl.max(axis=0))
    rp, rs = scans(zl.max(axis=1))
    # WHAT THE MARCH NEEDS of the entry it looks up for a sample in cell c (its f32 position's floor): a bound on every cell that
    # sample can read -- its bilinear corners floor(u), ceil(u) with u = s - 0.0001 lie in [c - 1, c + 1] (c - 1: an integral s),
    # and the f64 position the corners come from may sit one cell beside the f32 one: cells c - 2 ... c + 2 -- and on everything
    # further along the ray.  Stated on the TIGHT live set, without the kernel's construction.
    tcp, tcs = scans(zt.max(axis=0), back=2, ahead=2)
    trp, trs = scans(zt.max(axis=1), back=2, ahead=2)
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
    n_raw = (H * W + 16383) // 16384
    n_stat = (H * W + 32767) // 32768 if 8 < n_raw <= 16 else n_raw   # csrc/gcfr_march.hpp stat_chunk_px(): 512 x 512 has 8 records
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


def test_rays_running_along_column_zero_read_the_wrap_column_at_every_sample():
    """The case in which the wrap partner decides results (advisor r03): a light whose image-plane x is the image's left
    edge (C_x = -W/2 to within 1e-3), so that the rays of the pixels in column 0 climb straight up that column with
    u_x = -1e-4 ... 0 at every sample: each reads column W-1 as its left bilinear corner with weight ~1e-4.  The mask covers
    the left quarter only; column W-1 holds a masked-out wall of 1e5 ... 2e6 (z_A = z + 10 ... 200: it ramps with the height the
    climbing rays have gained, so their minima sit at LATE samples).  The trailing loop's cap must know that wall.  Grid schedule forced; against the C oracle, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import c_oracle
    from geomconsistentfr_amd import _lib, RenderParams, light_prep, shadow_min_distance
    dev = torch.device("cuda:0")
    H = W = 256
    prm = RenderParams()
    lights = []
    for ly, lz in ((0.8, 0.6), (0.9, 0.43), (0.6, 0.8), (0.95, 0.3)):
        a = -128.0 / 4013.0
        for _ in range(8):                                   # raw light whose prepared point has x = -128 (f32 arithmetic of T8:357-363)
            raw = np.array([[a, ly, lz]], np.float32)
            _, pt = c_oracle.light_prep(raw, clamp_z_min=0.0)
            a *= -128.0 / float(pt[0, 0])
        assert abs(float(pt[0, 0]) + 128.0) < 2e-3, pt
        lights.append(raw[0])
    lights = np.stack(lights)
    B = len(lights)
    rng = np.random.default_rng(5)
    r, c = np.mgrid[0:H, 0:W]
    depth = (8.0 + 3.0 * np.sin(r / 17.0) * np.cos(c / 13.0))[None].repeat(B, 0) + 0.2 * rng.random((B, H, W))
    depth = depth.astype(np.float32)
    mask = np.zeros((B, H, W), np.uint8)
    mask[:, :, :64] = 1                                      # touches column 0, nowhere near column W-1
    rows = np.arange(H)
    for b in range(B):                                       # the wall in the wrap partner column (masked out): 1e-4 x wall = 10 +
        slope = float(lights[b, 2] / lights[b, 1])           # half the height a ray from the bottom row has gained at that row, so
        depth[b, 2:200, W - 1] = (1e4 * (10.0 + 0.5 * slope * rows[::-1]))[2:200].astype(np.float32)   # that LATE samples win
    _, pt = light_prep(torch.from_numpy(lights).to(dev), prm)
    _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(prm.t0, prm.dt, prm.n_samples))
    assert (am_o[:, 0, :, 0] > 10).mean() > 0.8              # the wall decides: column 0's minima sit late on the rays
    lit = md_o < 1e5
    for want_argmin in (False, True):
        md, am = shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev), pt.reshape(B, 1, 3), prm,
                                     want_argmin=want_argmin, options=_lib.options(ksplit=0))
        bad = np.argwhere(md.cpu().numpy() != md_o)
        assert bad.size == 0, (want_argmin, len(bad), bad[:6])
        if want_argmin:
            assert np.array_equal(am.cpu().numpy()[lit], am_o[lit])


@pytest.mark.parametrize("edge", ["left", "top", "corner"])
def test_wrap_partner_of_an_edge_touching_mask_is_covered_by_the_tables(edge):
    """Advisor r03: a mask that touches column 0 (row 0) but not column W-1 (row H-1), and an extreme masked-out depth in that
    opposite column (row).  Samples whose rounded cell is in column 0 read column W-1 as their left bilinear corner (index -1
    wraps, T8:488-491; weight 1e-4 on the image's edge, up to 0.5 beside it): the horizon tables must contain those cells or the
    trailing loop's cap misses them.  Grid schedule forced, inference and argmin march, against the C oracle, bit for bit."""
    import os
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
    import c_oracle
    from geomconsistentfr_amd import _lib, RenderParams, light_prep, shadow_min_distance
    dev = torch.device("cuda:0")
    H = W = 256
    rng = np.random.default_rng(17)
    r, c = np.mgrid[0:H, 0:W]
    lights = np.array([[-0.9, 0.05, 0.3], [-0.6, 0.6, 0.5], [-0.3, -0.8, 0.4], [0.05, 0.9, 0.3], [0.4, 0.8, 0.4], [-0.7, 0.7, 0.1],
                       [-0.2, 0.3, 0.9], [0.0, 0.95, 0.2], [-0.95, 0.0, 0.2], [-0.5, 0.5, 0.7], [0.3, -0.2, 0.9], [-0.8, 0.5, 0.05]],
                      np.float32)
    B = len(lights)
    # a smooth surface that rises towards the touched edge (rays over it climb slowly: long trailing walks) + ripple
    base = 60.0 * np.exp(-((c / 90.0) ** 2 if edge != "top" else (r / 90.0) ** 2)) + 2.0 * np.sin(c / 9.0) * np.cos(r / 7.0)
    depth = (base[None] + 0.3 * rng.random((B, H, W))).astype(np.float32)
    mask = np.zeros((B, H, W), np.uint8)
    for b in range(B):
        if edge in ("left", "corner"):
            lo = 0 if edge == "corner" else 40 + 5 * b
            mask[b, lo:lo + 120, 0:60 + 3 * b] = 1            # touches column 0, far from column W-1
            depth[b, max(lo - 2, 0):lo + 122, W - 1] = (1e6, 3e4, 5e3)[b % 3]   # masked-out spike in the wrap partner column
        if edge in ("top", "corner"):
            lo = 0 if edge == "corner" else 30 + 6 * b
            mask[b, 0:50 + 2 * b, lo:lo + 130] = 1            # touches row 0, far from row H-1
            depth[b, H - 1, max(lo - 2, 0):lo + 132] = (1e6, 3e4, 5e3)[b % 3]
        if edge == "corner":
            depth[b, H - 1, W - 1] = 1e6
    prm = RenderParams()
    _, pt = light_prep(torch.from_numpy(lights).to(dev), prm)
    _, pt_o = c_oracle.light_prep(lights, clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o[:, None, :], c_oracle.sample_table(prm.t0, prm.dt, prm.n_samples))
    lit = md_o < 1e5
    assert lit.any()
    for want_argmin in (False, True):
        md, am = shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev), pt.reshape(B, 1, 3), prm,
                                     want_argmin=want_argmin, options=_lib.options(ksplit=0))
        bad = np.argwhere(md.cpu().numpy() != md_o)
        assert bad.size == 0, (want_argmin, len(bad), bad[:6])
        if want_argmin:
            assert np.array_equal(am.cpu().numpy()[lit], am_o[lit])


def test_config5_shape_with_forced_grid_schedule_against_the_c_oracle():
    """BASELINE configs[4]'s per-GPU shape -- one 512 x 512 face, elliptical mask, 18 lights, 320 samples -- through the grid
    schedule with trailing loop, horizon tables and octagon pruning (the shape where they gained most, +39 %), against the C
    oracle: minimum distance and argmin bit for bit."""
    import os
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "oracle"))
    sys.path.insert(0, root)
    import c_oracle
    import scenes
    from geomconsistentfr_amd import _lib, RenderParams, light_prep, shadow_min_distance
    dev = torch.device("cuda:0")
    depth, mask, _, _, light, _ = scenes.synth_faces_sized(1, 3, 512, 18, "ellipse")
    prm = RenderParams(n_samples=320, dt=0.8 / 320)
    _, pt = light_prep(torch.from_numpy(light.reshape(18, 3)).to(dev), prm)
    _, pt_o = c_oracle.light_prep(light.reshape(18, 3), clamp_z_min=0.0)
    md_o, am_o = c_oracle.shadow_min_distance(depth, mask, pt_o.reshape(1, 18, 3), c_oracle.sample_table(prm.t0, prm.dt, 320))
    lit = md_o < 1e5
    for want_argmin in (False, True):
        md, am = shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev), pt.reshape(1, 18, 3), prm,
                                     want_argmin=want_argmin, options=_lib.options(ksplit=0))
        assert np.array_equal(md.cpu().numpy(), md_o)
        if want_argmin:
            assert np.array_equal(am.cpu().numpy()[lit], am_o[lit])
