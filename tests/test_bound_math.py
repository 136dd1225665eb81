"""CPU check of the inequality behind the march kernel's depth-bound group skip.

The kernel skips a sample group when  (|G| - Kerr)^2 * 0.998 > bestS  for every candidate depth z of the group,
with  G = n (z - zb) - c1 t,  n = |BC_xy|,  c1 = BCz (d . BC_xy)/n.  That is only exact if the f32 value S the
body WOULD have computed (T8:497-509) is never below (|G| - Kerr)^2 * 0.998.  Here S is evaluated the reference's
way in float32 (positions in float64) for millions of random pixels / lights / samples / depths, and G and Kerr the
kernel's way in float32; the inequality must hold for every one of them.  No GPU, no library: numpy only.
"""
import numpy as np
import pytest

f32 = np.float32


def _case(rng, n, W, H, zscale, light_distance):
    x = (rng.integers(0, W, n) - W / 2).astype(f32)
    y = (H / 2 - rng.integers(0, H, n)).astype(f32)
    zb = (zscale * rng.standard_normal(n)).astype(f32)
    l = rng.standard_normal((n, 3))
    l[:, 2] = np.abs(l[:, 2]) * rng.choice([1.0, 0.05], n)
    l /= np.linalg.norm(l, axis=1, keepdims=True)
    C = (light_distance * l).astype(f32)
    # end point: towards the light's xy, somewhere short of the image border (any end point is fine for the bound:
    # the kernel takes d = E - B from its own end-point routine and only uses d . u)
    frac = rng.random(n).astype(f32)
    ang = rng.normal(0.0, 1e-3, n)                        # end points are not exactly on the ray to the light
    ux, uy = C[:, 0] - x, C[:, 1] - y
    rot_x = ux * np.cos(ang) - uy * np.sin(ang)
    rot_y = ux * np.sin(ang) + uy * np.cos(ang)
    scale = (frac * min(W, H)) / np.maximum(np.hypot(rot_x, rot_y), 1e-6)
    dx, dy = (rot_x * scale).astype(f32), (rot_y * scale).astype(f32)
    t = rng.uniform(0.025, 0.825, n)
    z = (zb + zscale * rng.standard_normal(n) * rng.choice([1.0, 0.01, 0.0], n)).astype(f32)   # the sampled depth
    return x, y, zb, C, dx, dy, t, z


@pytest.mark.parametrize("W,H,zscale,ld", [(256, 256, 40.0, 4013.0), (4096, 4096, 300.0, 4013.0), (64, 64, 0.01, 60.0),
                                           (512, 512, 1000.0, 1.0e5), (256, 256, 5.0, 500.0)])
def test_cross_product_distance_never_undercuts_the_bound(W, H, zscale, ld):
    rng = np.random.default_rng(W + int(zscale * 100))
    n = 400_000
    x, y, zb, C, dx, dy, t, z = _case(rng, n, W, H, zscale, ld)
    halfW, halfH = W / 2.0, H / 2.0
    # --- the body's S (reference arithmetic): f64 position pipeline, f32 distance (fma replaced by separately
    #     rounded products: one more rounding than the kernel, inside the error budget being tested)
    sx = x.astype(np.float64) + t * dx.astype(np.float64)
    sy = y.astype(np.float64) + t * dy.astype(np.float64)
    ux = (sx + halfW) - 0.0001
    uy = (halfH - sy) - 0.0001
    Ax, Ay, Az = (ux - halfW).astype(f32), (halfH - uy).astype(f32), z
    BAx, BAy, BAz = Ax - x, Ay - y, Az - zb
    BCx, BCy, BCz = C[:, 0] - x, C[:, 1] - y, C[:, 2] - zb
    Xx = BAy * BCz - BAz * BCy
    Xy = BAz * BCx - BAx * BCz
    Xz = BAx * BCy - BAy * BCx
    S = ((Xx * Xx + Xy * Xy) + Xz * Xz) + f32(1e-4)
    # --- the kernel's bound (float32, as in shadow_fwd_quad_kernel)
    nrm = np.sqrt(BCx * BCx + BCy * BCy)
    ok = nrm > 0
    c1 = BCz * ((dx * BCx + dy * BCy) / np.where(ok, nrm, f32(1)))
    Qz = nrm * zb
    tf = t.astype(f32)
    G = (nrm * z - Qz) - c1 * tf
    rr = np.maximum(np.maximum(np.abs(z - zb), np.abs(zb)), f32(max(H, W)))     # (the kernel uses the image-wide range: larger)
    K1 = f32(4e-3) * np.abs(BCz) + f32(1e-6) * np.abs(c1) * f32(0.825)
    K2 = f32(1e-6) * nrm + f32(2e-7) * ((np.abs(BCx) + np.abs(BCy)) + np.abs(BCz))
    Kerr = K2 * rr + K1 + nrm * (f32(1.2e-2) + f32(8e-6) * f32(max(H, W)))
    g = np.abs(G) - Kerr
    claim = ok & (g > 0)
    lower = (g * g * f32(0.998)).astype(np.float64)
    bad = claim & (S.astype(np.float64) < lower)
    assert claim.sum() > n // 10                      # the test must actually exercise the claim
    assert not bad.any(), (int(bad.sum()), float((lower[bad] / S[bad]).max()))


@pytest.mark.parametrize("kind", ["smooth", "noisy", "steep"])
def test_bilinear_samples_stay_inside_the_tiles_plane_band(kind):
    """The prepass bounds a tile by a band around a plane (slopes from quadrant means, clamped to +-4, band =
    residual extrema over the tile's cells).  A bilinear sample is a convex combination of four cells whose
    weighted mean position is the sample position, so it must lie in the same band AT that position."""
    rng = np.random.default_rng({"smooth": 1, "noisy": 2, "steep": 3}[kind])
    for _ in range(200):
        r, c = np.mgrid[0:16, 0:16].astype(np.float64)
        z = 30 * np.exp(-(((c - rng.uniform(0, 16)) / 9.0) ** 2 + ((r - rng.uniform(0, 16)) / 11.0) ** 2))
        if kind == "noisy":
            z = z + 5 * rng.random((16, 16))
        if kind == "steep":
            z = z + rng.uniform(-9, 9) * c + rng.uniform(-9, 9) * r        # beyond the +-4 clamp
        z = z.astype(f32)
        X0, Y0 = rng.integers(-128, 112), rng.integers(-111, 128)          # frame coordinates of cell (0, 0)
        X, Y = (X0 + c).astype(f32), (Y0 - r).astype(f32)                  # Y falls with the row
        m = [z[qy * 8:qy * 8 + 8, qx * 8:qx * 8 + 8].astype(f32).mean(dtype=f32) for qy in (0, 1) for qx in (0, 1)]
        a = np.clip(((m[1] + m[3]) - (m[0] + m[2])) * f32(0.0625), -4, 4).astype(f32)
        b = (-np.clip(((m[2] + m[3]) - (m[0] + m[1])) * f32(0.0625), -4, 4)).astype(f32)
        res = z - (a * X + b * Y)
        clo, chi = res.min(), res.max()
        u, v = rng.uniform(0, 15, 500), rng.uniform(0, 15, 500)            # sample positions (column, row)
        fu, fv = np.floor(u).astype(int), np.floor(v).astype(int)
        wx, wy = u - fu, v - fv
        zs = (z[fv, fu] * (1 - wx) + z[fv, fu + 1] * wx) * (1 - wy) + (z[fv + 1, fu] * (1 - wx) + z[fv + 1, fu + 1] * wx) * wy
        plane = a.astype(np.float64) * (X0 + u) + b.astype(np.float64) * (Y0 - v)
        tol = 1e-5 * (np.abs(zs) + np.abs(plane) + 1.0)                     # f32 roundings of the residuals
        assert np.all(zs - plane >= clo - tol) and np.all(zs - plane <= chi + tol)
