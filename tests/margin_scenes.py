"""Scene families for the march's safety margins (csrc/gcfr_mutants.hpp): inputs built from the geometry of each mechanism, free
parameters drawn from a seed.  Used by tools/mutant_hunt.py (which seeds make which mutant differ from the unmutated build) and by
tests/test_gpu_margins.py (the killing scenes, against the C oracle).  numpy only.

A scene is a dict: depth (B,H,W) f32, mask (B,H,W) u8, light_pt (B,L,3) f32 -- the light POINT, as gcfr_shadow_fwd takes it --,
t_table (N) f64; optional pixels_mask (also run with gcfr_options.pixels = 1).

Image-plane frame of the kernels: pixel (r, c) sits at x = c - W/2, y = H/2 - r; a ray runs from the pixel towards the light's
(x, y), clipped to the image box, and sample k sits at pixel + t_k (end - pixel)."""
import numpy as np

f32 = np.float32


def table(t0=0.025, dt=0.005, n=160):
    """gcfr_sample_table's rule (numpy's arange): t_k = t0 + k ((t0 + dt) - t0)"""
    delta = (t0 + dt) - t0
    return t0 + np.arange(n, dtype=np.float64) * delta


def grids(H, W):
    r, c = np.mgrid[0:H, 0:W]
    return (c - W / 2.0), (H / 2.0 - r), r, c


def far_light(az, el, R):
    return np.array([R * np.cos(el) * np.cos(az), R * np.cos(el) * np.sin(az), R * np.sin(el)], np.float64)


# ---------------------------------------------------------------------------------------------------------------------------
# A  rays PARALLEL to a plane, a hair above / below it.  A far light makes the rays parallel (direction u in the image plane,
#    slope tan(el)); the surface z = tan(el) (u . p) + c contains them.  The pixels of interest lie in a masked-OUT half-plane
#    whose surface is the same plane shifted by eps(v): their first unmasked sample already lies on the plane, and from there
#    on EVERY sample has the same G = n eps up to rounding -- every group is a near-tie of the running minimum, so whether a
#    group may be skipped is decided by the bound's error terms (Kerr, the 0.2 % slack) and nothing else.  Bounds tiles away
#    from the step are exact planes: thin bands, tight bounds.
# ---------------------------------------------------------------------------------------------------------------------------
def family_parallel(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(1000 + seed)
    X, Y, _, _ = grids(H, W)
    depth, mask, lights = [], [], []
    R = float(rng.choice([4013.0, 1e5, 1e6, 1e7]))
    base = float(rng.choice([0.0, 0.0, 150.0, 3.0e4, 1.0e5]))
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        el = float(rng.choice([rng.uniform(2e-4, 0.02), rng.uniform(0.02, 0.5), rng.uniform(0.5, 1.3)]))
        C = far_light(az, el, R)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]                                  # position along the rays
        v = -X * u[1] + Y * u[0]                                 # ... across them
        alpha = min(np.tan(el), 3.9)                              # (the plane fit clamps its slopes to +-4)
        plane = alpha * s + base
        eps0 = 10.0 ** rng.uniform(-2.6, 0.6) * float(rng.choice([-1.0, 1.0]))
        eps = eps0 * (1.0 + 0.25 * v / max(H, W))
        s0 = rng.uniform(-0.25, 0.1) * min(H, W)
        behind = s < s0
        depth.append(np.where(behind, plane - eps, plane).astype(f32))
        mask.append((~behind).astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# C  distances around the masked value 1e6 (T8:512): a plateau so high above the pixels that sqrt(S) / |BC| lands within a few
#    per cent of 1e6, under masks with holes -- `any_masked` decides the result exactly when the distance is not < 1e6, so every
#    shortcut that stops looking at the mask ("bestS < safeS", the trailing loop, lanes that have left the box) is on the line.
# ---------------------------------------------------------------------------------------------------------------------------
def family_million(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(3000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        el = rng.uniform(0.05, 1.2)
        az = rng.uniform(0, 2 * np.pi)
        C = far_light(az, el, float(rng.choice([4013.0, 1e5])))
        # distance of a point dz above the ray's foot from the ray ~ dz cos(el): aim at 1e6 (0.97 ... 1.03)
        dz = 1.0e6 / np.cos(el) * rng.uniform(0.97, 1.03)
        low = (X * np.cos(az) + Y * np.sin(az)) < rng.uniform(-0.3, 0.2) * W      # the pixels: low ground, part of it unmasked
        z = np.where(low, 0.0, dz) + rng.uniform(0, 50) * rng.random((H, W)) * rng.choice([0.0, 1.0])
        m = np.ones((H, W), bool)
        kind = rng.integers(0, 4)
        if kind == 0:
            m = ~low                                             # only the plateau is inside the mask
        elif kind == 1:
            m = ~low | (rng.random((H, W)) < 0.3)
        elif kind == 2:
            m = (np.abs(X) < 0.3 * W) & (np.abs(Y) < 0.35 * H)
        else:
            m = rng.random((H, W)) < 0.97                        # a few holes
        depth.append(z.astype(f32))
        mask.append(m.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# D  DESCENDING rays: pixels on a high, masked-out plateau, the light below them (BCz < 0, c1 < 0), the masked-in surface lower
#    still.  Early on the ray is far above everything it can sample -- the termination bound holds NOW -- but it comes down onto
#    the surface later: only "c1 > 0" (the ray is still rising) makes the bound monotone.
# ---------------------------------------------------------------------------------------------------------------------------
def family_descending(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(4000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        top = rng.uniform(60, 400)
        high = s < rng.uniform(-0.35, -0.1) * W
        ground = rng.uniform(0, 5) + 0.02 * s * rng.choice([0.0, 1.0])
        z = np.where(high, top, ground)
        # the light: ahead of the pixels, BELOW the plateau, a little above the ground
        dist = rng.uniform(0.8, 3.0) * W
        C = np.array([u[0] * dist, u[1] * dist, rng.uniform(0.0, 0.5) * top])
        depth.append(z.astype(f32))
        mask.append((~high).astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# E  DIAMOND masks against rays that run along their edges: a sample whose rounded cell is a mask cell on the diagonal edge can
#    sit up to 1 outside the edge in x -+ y (two roundings of 0.5) -- the octagon's inflation.  Few samples of such a ray are
#    unmasked, so losing one changes the minimum.
# ---------------------------------------------------------------------------------------------------------------------------
def family_diamond(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(5000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        cx, cy = rng.integers(-W // 6, W // 6), rng.integers(-H // 6, H // 6)
        rad = int(rng.integers(3, W // 4))
        m = (np.abs(X - cx) + np.abs(Y - cy)) <= rad
        az = np.pi / 4 * (1 + 2 * rng.integers(0, 4)) + rng.normal(0, 0.02)
        C = far_light(az, rng.uniform(0.2, 1.0), float(rng.choice([4013.0, 1e6])))
        z = 20 * rng.random((H, W))
        depth.append(z.astype(f32))
        mask.append(m.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=True)


# ---------------------------------------------------------------------------------------------------------------------------
# F  sample tables at the edge of what the prepass accepts (increasing, inside [0, 1], every step within 0.1 % of the mean step):
#    the first half of the steps 0.09 % long, the second half 0.09 % short -- sample k sits up to 0.07 steps away from where a
#    uniform table has it, so the conversion of a ray's [t_in, t_out] into sample indices needs its slack.  Small masks: few
#    samples per ray are unmasked, the first / last of them decide the minimum.
# ---------------------------------------------------------------------------------------------------------------------------
def family_table(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(6000 + seed)
    X, Y, r, c = grids(H, W)
    n = 160
    steps = np.full(n - 1, 0.005)
    sign = rng.choice([-1.0, 1.0])
    steps[:(n - 1) // 2] *= 1.0 + sign * 0.0009
    steps[(n - 1) // 2:] *= 1.0 - sign * 0.0009
    tt = 0.025 + np.concatenate([[0.0], np.cumsum(steps)])
    depth, mask, lights = [], [], []
    for b in range(B):
        m = np.zeros((H, W), bool)
        for _ in range(rng.integers(1, 4)):
            r0, c0 = rng.integers(0, H - 4), rng.integers(0, W - 4)
            m[r0:r0 + rng.integers(1, 4), c0:c0 + rng.integers(1, 4)] = True
        z = 30 * rng.random((H, W)) * rng.choice([0.0, 1.0]) + rng.uniform(0, 10)
        C = far_light(rng.uniform(0, 2 * np.pi), rng.uniform(0.05, 1.2), 4013.0)
        depth.append(z.astype(f32))
        mask.append(m.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=tt, pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# I  cliffs under a COARSE table: 95 samples 0.0085 apart move a ray up to 6.5 cells per group of four at 256 px -- a footprint of
#    ten cells with its bilinear corners, more than a depth-bounds tile of stride 8 covers from the cell that selects it.  The
#    prepass picks the stride from the table (a group's reach + 3 cells); piecewise-constant terraces with tall cliffs make a
#    bound taken from a tile that does not cover the cliff wrong.
# ---------------------------------------------------------------------------------------------------------------------------
def family_cliffs(seed, H=256, W=256, B=4):
    rng = np.random.default_rng(8000 + seed)
    X, Y, r, c = grids(H, W)
    depth, lights = [], []
    for b in range(B):
        cell = int(rng.choice([8, 8, 16, 5]))
        ph_r, ph_c = rng.integers(0, cell), rng.integers(0, cell)
        levels = rng.uniform(0, 60, ((H + cell) // cell + 2, (W + cell) // cell + 2))
        z = levels[(r + ph_r) // cell, (c + ph_c) // cell]
        C = far_light(rng.uniform(0, 2 * np.pi), rng.uniform(0.02, 0.6), 4013.0)
        depth.append(z.astype(f32))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.ones((B, H, W), np.uint8), light_pt=np.array(lights, f32),
                t_table=table(0.025, 0.0085, 95), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# J  surfaces BELOW zero under grazing rays: the termination cap is max(depth maximum, 0) -- the 0 because a sample at an
#    integral coordinate reads z = 0 whatever the depth map holds -- and the rough candidates for "without the 0" are images
#    whose depth is negative everywhere, marched WITHOUT horizon tables (W % 4 != 0: the tables need whole dwords of columns).
# ---------------------------------------------------------------------------------------------------------------------------
def family_negative(seed, H=128, W=130, B=8):
    rng = np.random.default_rng(9000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        z = -rng.uniform(20, 200) - rng.uniform(0, 30) * rng.random((H, W)) * rng.choice([0.0, 1.0])
        az = rng.uniform(0, 2 * np.pi)
        C = far_light(az, rng.uniform(0.01, 0.4), 4013.0)
        C[2] = rng.uniform(-150, 0)                                # the light below zero as well: the rays stay under z = 0
        depth.append(z.astype(f32))
        mask.append((rng.random((H, W)) < rng.choice([1.0, 0.8])).astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


FAMILIES = {"parallel": family_parallel, "million": family_million, "descending": family_descending,
            "diamond": family_diamond, "table": family_table, "cliffs": family_cliffs, "negative": family_negative}


# ---------------------------------------------------------------------------------------------------------------------------
# H  a sample at an INTEGRAL unrounded coordinate.  The reference's bilinear weights are ceil(u) - u and u - floor(u) (T8:492-494):
#    both are 0 when u is integral, and the sampled depth is 0 whatever the depth map holds.  u_x = (s_x + W/2) - 0.0001 is
#    integral only if t_k dx equals 0.0001 to the last bit of an f64 -- found here by search: the pixel in column W/2 (x = 0, so
#    that dx = Cx is any f32 we like), the light's x a few 1e-4 px beside it, over the samples k of a table and the f32
#    neighbours of 0.0001 / t_k.  The depth map is flat and NEGATIVE, the light's height makes the ray cross z = 0 exactly at
#    sample k: that sample is the minimum by far, and nothing in the depth map (the bounds tiles, the depth maximum) knows of it.
# ---------------------------------------------------------------------------------------------------------------------------
def integral_hits(W, t0, dt=0.005, n=160, span=40):
    """[(k, dx as f32)] with ((0 + t_k dx) + W/2) - 0.0001 integral in f64, dx > 0"""
    tt = table(t0, dt, n)
    hits = []
    halfW = W / 2.0
    for k in range(n):
        d0 = f32(0.0001 / tt[k])
        cand = np.array([d0], f32)
        lo, hi = d0, d0
        for _ in range(span):
            lo, hi = np.nextafter(lo, f32(0)), np.nextafter(hi, f32(1))
            cand = np.concatenate([cand, [lo, hi]])
        sx = 0.0 + tt[k] * cand.astype(np.float64)
        ux = (sx + halfW) - 0.0001
        for dx in cand[ux == np.floor(ux)]:
            hits.append((k, f32(dx)))
    return tt, hits


def family_integral(seed, H=64, B=4):
    rng = np.random.default_rng(10000 + seed)
    W = 64 if seed % 2 == 0 else 66            # 66: W % 4 != 0 -- no horizon tables, the termination cap is the image-wide one
    scenes_tt, hits = None, []
    t0 = 0.025
    for trial in range(400):
        t0 = 0.02 + 0.0001 * ((seed * 400 + trial) % 3000) + 1e-6 * rng.integers(0, 90)
        scenes_tt, hits = integral_hits(W, t0)
        hits = [h for h in hits if 8 <= h[0] < 150]
        if hits:
            break
    depth, lights = [], []
    zb = -40.0
    for b in range(B):
        k, dx = hits[b % len(hits)] if hits else (20, f32(1e-3))
        r0 = 40 + b                                                   # the pixel: (r0, W/2): x = 0, y = H/2 - r0
        y = H / 2.0 - r0
        z = np.full((H, W), zb) - (0.5 * rng.random((H, W)) if b % 2 else 0.0)
        z[r0, W // 2] = zb
        tk = scenes_tt[k]
        depth.append(z.astype(f32))
        lights.append([[float(dx), y + 25.0, zb + 40.0 / tk]])        # the ray is at height 0 at t_k
    return dict(depth=np.stack(depth), mask=np.ones((B, H, W), np.uint8), light_pt=np.array(lights, f32), t_table=scenes_tt,
                pixels_mask=False, critical=[(b, 40 + b, W // 2) for b in range(B)], found=len(hits))


FAMILIES["integral"] = family_integral


# ---------------------------------------------------------------------------------------------------------------------------
# K  LEVEL rays high above a flat surface.  The pixels lie on a masked-out plateau `top` above the masked-in ground (a masked-out
#    strip of ground between them keeps the plateau out of the live cells the termination cap is taken over); the light is level
#    with the plateau, a hair above it: c1 > 0 but tiny, every sample is the same distance n * top from the ray up to rounding --
#    the minimum is wherever the f32 noise puts it, often late.  The termination test sees a ray `top` above the cap, i.e. a bound
#    only Kerr (relative 1e-4 at this height) below the running minimum: the 0.2 % slack is what keeps it from stopping early.
# ---------------------------------------------------------------------------------------------------------------------------
def family_level(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(11000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        top = 10.0 ** rng.uniform(1.3, 3.0)
        edge = rng.uniform(-0.3, -0.1) * W
        high = s < edge
        live = s > edge + 12                                         # masked-in ground starts 12 px behind the plateau's edge
        z = np.where(high, top, 0.0)
        dist = 10.0 ** rng.uniform(2.5, 5.0)
        C = np.array([u[0] * dist, u[1] * dist, top * (1.0 + 10.0 ** rng.uniform(-7, -3))])
        depth.append(z.astype(f32))
        mask.append(live.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# L  TWO far plateaus around the masked value.  Seen from the low ground, plateau A is 1e6 (0.99 ... 1.02) away from the rays,
#    plateau B, further along them, some per cent more: over B the depth bound says "cannot win" by a wide margin, so the wave
#    would walk those groups without looking at the mask (the trailing loop) -- which is only allowed for a lane whose minimum is
#    CERTAINLY below 1e6 (bestS < safeS), because for the others a masked sample -- the holes in B's mask -- decides the result.
# ---------------------------------------------------------------------------------------------------------------------------
def family_million2(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(12000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        el = rng.uniform(0.05, 1.0)
        az = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        C = far_light(az, el, float(rng.choice([4013.0, 1e5])))
        dzA = 1.0e6 / np.cos(el) * rng.uniform(0.985, 1.025)
        dzB = dzA * rng.uniform(1.02, 1.3)
        e1 = rng.uniform(-0.35, -0.05) * W
        e2 = e1 + rng.uniform(0.1, 0.4) * W
        z = np.where(s < e1, 0.0, np.where(s < e2, dzA, dzB))
        m = np.ones((H, W), bool)
        holes = (s >= e2) & (rng.random((H, W)) < rng.choice([0.01, 0.05, 0.3]))
        m[holes] = False
        if rng.random() < 0.5:
            m[s < e1] = rng.random() < 0.5                           # the pixels themselves inside or outside the mask
        depth.append(z.astype(f32))
        mask.append(m.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# M  cliffs under a table that moves a ray 7.9 cells per group of four at 256 px: every ray longer than 227 px spans eight
#    rounded cells within a group, one more than a depth-bounds tile of stride 8 covers from the cell that selects it.  Lights
#    along the axes (long rays), terraces of 8 x 8 cells at every phase against the tile grid.
# ---------------------------------------------------------------------------------------------------------------------------
def family_cliffs2(seed, H=256, W=256, B=4):
    rng = np.random.default_rng(13000 + seed)
    X, Y, r, c = grids(H, W)
    depth, lights = [], []
    for b in range(B):
        cell = int(rng.choice([8, 8, 4, 16]))
        ph_r, ph_c = rng.integers(0, cell), rng.integers(0, cell)
        levels = rng.uniform(0, 80, ((H + cell) // cell + 2, (W + cell) // cell + 2)) * rng.choice([1.0, 1.0, 0.2])
        z = levels[(r + ph_r) // cell, (c + ph_c) // cell]
        az = np.pi / 2 * rng.integers(0, 4) + rng.normal(0, 0.12)
        C = far_light(az, rng.uniform(0.02, 0.5), 4013.0)
        depth.append(z.astype(f32))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.ones((B, H, W), np.uint8), light_pt=np.array(lights, f32),
                t_table=table(0.025, 0.0103, 78), pixels_mask=False)


FAMILIES.update({"level": family_level, "million2": family_million2, "cliffs2": family_cliffs2})


# ---------------------------------------------------------------------------------------------------------------------------
# A2 the parallel-plane construction of family A in the three regimes where ONE term of Kerr = K1 + K2 r + n (...) carries the
#    bound: (0) steep rays under a near light, K1 = 4e-3 |BCz| as large as the plane term; (1) depth values around 1e5 -- r, the
#    bound on |BA|'s components, is 1e5 and K2 r = (1e-6 n + 2e-7 |BC|_1) r dwarfs the rest (and the f32 products of the distance
#    really are that noisy there); (2) grazing far lights over a steep plane: only the plane-evaluation term is left.
# ---------------------------------------------------------------------------------------------------------------------------
def family_parallel2(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(14000 + seed)
    regime = seed % 3
    X, Y, _, _ = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        if regime == 0:
            el, R, base = rng.uniform(1.15, 1.32), 4013.0, 0.0
        elif regime == 1:
            el, R, base = rng.uniform(0.05, 1.2), 4013.0, float(rng.choice([3.0e4, 1.0e5, 3.0e5]))
        else:
            el, R, base = rng.uniform(0.3, 1.3), float(rng.choice([1e5, 1e6])), 0.0
        C = far_light(az, el, R)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        v = -X * u[1] + Y * u[0]
        alpha = min(np.tan(el), 3.9)
        plane = alpha * s + base
        lo = -1.0 if regime == 1 else -2.3
        eps0 = 10.0 ** rng.uniform(lo, 0.8) * float(rng.choice([-1.0, 1.0]))
        eps = eps0 * (1.0 + 0.25 * v / max(H, W))
        s0 = rng.uniform(-0.25, 0.1) * min(H, W)
        behind = s < s0
        depth.append(np.where(behind, plane - eps, plane).astype(f32))
        mask.append((~behind).astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


FAMILIES["parallel2"] = family_parallel2


# ---------------------------------------------------------------------------------------------------------------------------
# K2 level rays over ground that comes UP towards them by less than the termination tests' slack.  As family K (pixels on a masked-out
#    plateau `top` above the ground, the light level with them), but the ground carries (a) a ramp that rises by 0.01 ... 0.09 % of
#    `top` over the ray, or (b) a bump h1 behind the plateau's edge, a dip, and a far bump h2 = h1 + 0.01 ... 0.09 % of `top`:
#    the far samples are CLOSER to the ray than the running minimum, by less than 0.2 % -- a termination bound without its slack
#    (over the image-wide cap in the main loop, over the horizon tables' cap in the trailing loop the dip puts the wave into)
#    calls the march off in front of them.
# ---------------------------------------------------------------------------------------------------------------------------
def family_level2(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(15000 + seed)
    X, Y, r, c = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        top = 10.0 ** rng.uniform(1.6, 3.0)
        edge = rng.uniform(-0.35, -0.15) * W
        high = s < edge
        live = s > edge + 12
        rise = top * rng.uniform(1e-4, 9e-4)
        if (seed + b) % 2 == 0:                                       # (a) a ramp
            ground = np.clip((s - edge - 12) / (0.6 * W), 0.0, 1.0) * rise
        else:                                                         # (b) bump, dip, far bump
            h1 = top * rng.uniform(0.003, 0.02)
            s2 = edge + rng.uniform(45, 70)
            ground = np.where(s < edge + 26, h1, np.where(s < s2, 0.0, h1 + rise))
        z = np.where(high, top, ground)
        dist = 10.0 ** rng.uniform(2.5, 4.5)
        C = np.array([u[0] * dist, u[1] * dist, top * (1.0 + 10.0 ** rng.uniform(-6.5, -4))])
        depth.append(z.astype(f32))
        mask.append(live.astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# L2 family L lifted far above zero.  Every bound also has to hold for the isolated sampled value z = 0 (integral coordinates), so
#    near z = 0 no group is ever "certainly lost" and the wave never walks without its mask; with everything 3e6 ... 6e6 up, the two
#    plateaus around the masked value do put it there.
# ---------------------------------------------------------------------------------------------------------------------------
def family_million3(seed, H=128, W=128, B=8):
    sc = family_million2(seed, H, W, B)
    rng = np.random.default_rng(16000 + seed)
    base = rng.uniform(3e6, 6e6, (B, 1, 1))
    sc["depth"] = (sc["depth"].astype(np.float64) + base).astype(f32)
    sc["light_pt"] = sc["light_pt"].copy()
    sc["light_pt"][:, :, 2] += base[:, :, 0].astype(f32)
    return sc


FAMILIES.update({"level2": family_level2, "million3": family_million3})


# ---------------------------------------------------------------------------------------------------------------------------
# A3 the parallel-plane construction with the whole scene 1e6 ... 1e7 above zero: the f32 products of the distance (|BA| |BC| 1.2e-7)
#    are then hundreds of units of G, far more than K1 and the plane term together -- K2 r, the term that grows with the depth
#    range, is the only thing between that noise and a wrong skip.
# ---------------------------------------------------------------------------------------------------------------------------
def family_parallel3(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(17000 + seed)
    X, Y, _, _ = grids(H, W)
    depth, mask, lights = [], [], []
    for b in range(B):
        az = rng.uniform(0, 2 * np.pi)
        el = rng.uniform(0.05, 1.0)
        base = float(rng.choice([1.0e6, 3.0e6, 1.0e7]))
        C = far_light(az, el, 4013.0)
        C[2] += base
        u = np.array([np.cos(az), np.sin(az)])
        s = X * u[0] + Y * u[1]
        v = -X * u[1] + Y * u[0]
        plane = np.tan(el) * s + base
        eps0 = 10.0 ** rng.uniform(-0.3, 2.2) * float(rng.choice([-1.0, 1.0]))
        eps = eps0 * (1.0 + 0.25 * v / max(H, W))
        s0 = rng.uniform(-0.25, 0.1) * min(H, W)
        behind = s < s0
        depth.append(np.where(behind, plane - eps, plane).astype(f32))
        mask.append((~behind).astype(np.uint8))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.stack(mask), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# N  steep planar FACETS under grazing light.  32 x 32-pixel facets, each an exact plane with slopes up to +-3.9 per axis (what the
#    bounds tiles' plane fit represents exactly: thin bands, tight bounds); the light far away and almost level, so K1 ~ |BCz| is
#    nothing.  The sample POSITION the reference evaluates is 1e-4 beside s(t) (T8:480-487): on a slope of 4 + 4 the sampled depth
#    differs by 8e-4 from the plane's value at s(t) -- the plane-evaluation term of Kerr.  G runs through the facets linearly, five
#    n per sample: among thousands of pixels some have a later group's closest sample within that 8e-4 n of their running minimum.
# ---------------------------------------------------------------------------------------------------------------------------
def family_facets(seed, H=128, W=128, B=8):
    rng = np.random.default_rng(18000 + seed)
    X, Y, r, c = grids(H, W)
    depth, lights = [], []
    for b in range(B):
        cell = 32
        ph_r, ph_c = 8 * rng.integers(0, 4), 8 * rng.integers(0, 4)
        fi, fj = (r + ph_r) // cell, (c + ph_c) // cell
        nf = (H // cell + 2, W // cell + 2)
        a = rng.uniform(-3.9, 3.9, nf)[fi, fj]
        bb = rng.uniform(-3.9, 3.9, nf)[fi, fj]
        off = rng.uniform(-30, 30, nf)[fi, fj]
        z = a * X + bb * Y + off
        C = far_light(rng.uniform(0, 2 * np.pi), rng.uniform(1e-3, 0.1), float(rng.choice([4013.0, 1e5])))
        depth.append(z.astype(f32))
        lights.append([C])
    return dict(depth=np.stack(depth), mask=np.ones((B, H, W), np.uint8), light_pt=np.array(lights, f32), t_table=table(), pixels_mask=False)


# ---------------------------------------------------------------------------------------------------------------------------
# B2 pits in a plateau under an overhead light, the plateau's height SWEPT in steps of 0.004 through the value at which sample 7
#    (below the plateau) and sample 8 (above it: the first sample of the group behind a termination check) are equally far from the
#    ray.  Near the light's foot n is 5 ... 40 px and G = n (P - BCz t): the reference's 1e-4 position offset moves G by 1.4e-4 |BCz|
#    = 0.56 there, more than the plane term (0.013 n) and K2 r together -- K1 = 4e-3 |BCz| is the term that covers it.  One pit per
#    tile of the march (16 x 4), so that a tile's fate hangs on one lane.
# ---------------------------------------------------------------------------------------------------------------------------
def family_pits2(seed, H=128, W=128, B=24):
    rng = np.random.default_rng(19000 + seed)
    Cz = 4013.0
    tt = table()
    Cx, Cy = rng.uniform(-0.25, 0.25) * W, rng.uniform(-0.25, 0.25) * H
    j = int(rng.choice([7, 15]))
    mid = Cz * 0.5 * (tt[j] + tt[j + 1])
    _, _, r, c = grids(H, W)
    pit = ((r % 4) == rng.integers(0, 4)) & ((c % 16) == rng.integers(0, 16))
    depth, lights = [], []
    for b in range(B):
        P = mid + (b - 0.7 * B) * 0.004
        depth.append(np.where(pit, 0.0, P).astype(f32))
        lights.append([[Cx + 0.37, Cy - 0.21, Cz]])
    return dict(depth=np.stack(depth), mask=np.ones((B, H, W), np.uint8), light_pt=np.array(lights, f32), t_table=tt, pixels_mask=False)


FAMILIES.update({"parallel3": family_parallel3, "facets": family_facets, "pits2": family_pits2})


# ---------------------------------------------------------------------------------------------------------------------------
# W  a mask that touches column 0 (row 0) only, an extreme masked-OUT depth in the opposite column (row): samples whose rounded cell
#    lies in column 0 read column W-1 as their left bilinear corner (index -1 wraps, T8:488-491), so the horizon tables' entries must
#    hold those cells or the trailing loop's cap is too low (advisor r03; tests/test_gpu_horizon.py pins the product end to end with
#    these scenes, tools/audit.py audits the terminations' claims on them: mutants 13 / 14).  seed % 3: left / top / corner.
# ---------------------------------------------------------------------------------------------------------------------------
def family_wrap_edge(seed, H=256, W=256, B=12):
    edge = ("left", "top", "corner")[seed % 3]
    rng = np.random.default_rng(23000 + seed)
    _, _, r, c = grids(H, W)
    raw = np.array([[-0.9, 0.05, 0.3], [-0.6, 0.6, 0.5], [-0.3, -0.8, 0.4], [0.05, 0.9, 0.3], [0.4, 0.8, 0.4], [-0.7, 0.7, 0.1],
                    [-0.2, 0.3, 0.9], [0.0, 0.95, 0.2], [-0.95, 0.0, 0.2], [-0.5, 0.5, 0.7], [0.3, -0.2, 0.9], [-0.8, 0.5, 0.05]], np.float64)
    raw = raw + 0.02 * rng.standard_normal(raw.shape)
    raw[:, 2] = np.abs(raw[:, 2])
    pts = (4013.0 * raw / np.linalg.norm(raw, axis=1, keepdims=True)).astype(f32)[:B]
    base = 60.0 * np.exp(-((c / 90.0) ** 2 if edge != "top" else (r / 90.0) ** 2)) + 2.0 * np.sin(c / 9.0) * np.cos(r / 7.0)
    depth = (base[None] + 0.3 * rng.random((B, H, W))).astype(f32)
    mask = np.zeros((B, H, W), np.uint8)
    for b in range(B):
        spike = (1e6, 3e4, 5e3)[(b + seed // 3) % 3]
        if edge in ("left", "corner"):
            lo = 0 if edge == "corner" else 40 + 5 * b
            mask[b, lo:lo + 120, 0:60 + 3 * b] = 1
            depth[b, max(lo - 2, 0):lo + 122, W - 1] = spike
        if edge in ("top", "corner"):
            lo = 0 if edge == "corner" else 30 + 6 * b
            mask[b, 0:50 + 2 * b, lo:lo + 130] = 1
            depth[b, H - 1, max(lo - 2, 0):lo + 132] = spike
        if edge == "corner":
            depth[b, H - 1, W - 1] = 1e6
    return dict(depth=depth, mask=mask, light_pt=pts[:, None, :], t_table=table(), pixels_mask=False)


FAMILIES["wrap_edge"] = family_wrap_edge


# ---------------------------------------------------------------------------------------------------------------------------
# W2 the wrap partner DECIDES results and late samples win (tests/test_gpu_horizon.py::test_rays_running_along_column_zero...,
#    round 6: into the audit's list).  The light point's x is the image's left edge EXACTLY (C_x = -W/2: the reference's nine-way
#    branch still treats it as "inside", T8:404), so the rays of column 0 climb straight up that column with u_x = -1e-4 at every
#    sample and read column W-1 as their left bilinear corner with weight 1e-4 (index -1 wraps, T8:488-491).  Column W-1 is masked
#    out and holds a wall that RAMPS with the height a climbing ray has gained (1e-4 x wall ~ the ray's own height: the blended
#    surface stays near the ray, so the minima of column 0 sit at LATE samples, inside the trailing loop's reach).  A horizon
#    table that leaves the wrap partner out (mutant 13) or a dilation that does not wrap (mutant 14) puts the cap below that wall:
#    the termination's claim `S_k >= 0.998 g^2` is then false for samples that were never evaluated -- what tools/audit.py counts.
#    seed % 2: the left edge (columns) / the top edge (rows: C_y = H/2, rays running along row 0 read row H-1).
# ---------------------------------------------------------------------------------------------------------------------------
def family_wrap_column(seed, H=256, W=256, B=4):
    rng = np.random.default_rng(29000 + seed)
    _, _, r, c = grids(H, W)
    top = seed % 2 == 1
    dirs = [(0.8, 0.6), (0.9, 0.43), (0.6, 0.8), (0.95, 0.3)][:B]
    pts, depth = [], []
    mask = np.zeros((B, H, W), np.uint8)
    for b, (along, lz) in enumerate(dirs):
        along, lz = along + 0.03 * rng.standard_normal(), lz + 0.03 * rng.standard_normal()
        n = np.hypot(along, lz)
        R = 4013.0
        far, up = R * along / n, R * abs(lz) / n
        d = (8.0 + 3.0 * np.sin(r / 17.0) * np.cos(c / 13.0) + 0.2 * rng.random((H, W))).astype(f32)
        slope = up / far                                     # height gained per pixel travelled along the edge
        if not top:                                          # light at x = -W/2 exactly, above the image: rays of column 0 climb the rows
            pts.append([-(W / 2.0), far, up])
            mask[b, :, :64] = 1                              # touches column 0, nowhere near column W-1
            gained = 0.5 * slope * np.arange(H)[::-1]        # half the height a ray from the bottom row has gained at that row
            d[2:200, W - 1] = (1e4 * (10.0 + gained))[2:200].astype(f32)
        else:                                                # light at y = H/2 exactly, to the right: rays of row 0 run along the columns
            pts.append([far, H / 2.0, up])
            mask[b, :64, :] = 1                              # touches row 0, nowhere near row H-1
            gained = 0.5 * slope * np.arange(W)
            d[H - 1, 2:200] = (1e4 * (10.0 + gained))[2:200].astype(f32)
        depth.append(d)
    return dict(depth=np.stack(depth), mask=mask, light_pt=np.array(pts, f32)[:, None, :], t_table=table(), pixels_mask=False)


FAMILIES["wrap_column"] = family_wrap_column


# ---------------------------------------------------------------------------------------------------------------------------
# W3 the PREFIX tables' wrap partner (mutant 13): a sample table that reaches t = 1.  With the reference's tables (t <= 0.82) a ray
#    heading towards column 0 never gets there (csrc/gcfr_mutants.hpp, note on 13 / 14); a caller's table may end at t = 1 -- the
#    prepass accepts tables inside [0, 1] -- and then the LAST sample of a ray that ends on the image's left edge sits at
#    u_x = -1e-4: its left bilinear corner is column W-1 (index -1 wraps, T8:488-491), weight 1e-4.  Light far to the left and level
#    (every ray ends on x = -W/2, T8:399-403), mask on the left quarter, column W-1 masked out and holding, per row, the wall that
#    lifts the blended surface of that row's column-0 cell to the height the ray of the pixel K0 columns in has reached at t = 1:
#    for pixels around column K0 the LAST sample is the minimum.  The trailing loop's cap for a ray heading left is the prefix
#    maximum over the columns it can still touch -- which must hold column W-1.
# ---------------------------------------------------------------------------------------------------------------------------
def family_wrap_last_sample(seed, H=256, W=256, B=4, K0=30):
    rng = np.random.default_rng(31000 + seed)
    X, Y, r, c = grids(H, W)
    n = 160
    tt = 0.2 + np.arange(n, dtype=np.float64) * ((0.8 / (n - 1)) * (1.0 - 1e-9))        # t_159 = 1 - 8e-10: inside [0, 1]
    depth, pts = [], []
    mask = np.zeros((B, H, W), np.uint8)
    mask[:, :, :64] = 1
    for b in range(B):
        el = np.deg2rad(rng.uniform(22.0, 40.0))
        az = np.pi + np.deg2rad(rng.uniform(-0.6, 0.6))
        C = far_light(az, el, 4013.0)
        d = (8.0 + 1.5 * np.sin(r / 19.0) * np.cos(c / 15.0) + 0.1 * rng.random((H, W))).astype(f32)
        k0 = K0 + int(rng.integers(-6, 7))
        x0, y0, z0 = X[:, k0], Y[:, k0], d[:, k0].astype(np.float64)
        s_end = (-(W / 2.0) - x0) / (C[0] - x0)                   # the ray's 3-D parameter where it crosses x = -W/2
        h = z0 + s_end * (C[2] - z0)                              # ... and its height there
        y_end = y0 + s_end * (C[1] - y0)
        rows = np.clip(np.rint(H / 2.0 - y_end).astype(int), 0, H - 1)   # the row whose column-0 / column-(W-1) cells that sample blends
        wall = np.full(H, 9.0)
        wall[rows] = (h - (1.0 - 1e-4) * d[rows, 0].astype(np.float64)) / 1e-4
        d[:, W - 1] = wall.astype(f32)
        depth.append(d)
        pts.append([C])
    return dict(depth=np.stack(depth), mask=mask, light_pt=np.array(pts, f32), t_table=tt, pixels_mask=False)


FAMILIES["wrap_last_sample"] = family_wrap_last_sample
