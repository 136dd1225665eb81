"""GPU: analytic pins of the normals stage (T8:353-354; kornia 0.4.1 restated -- bit parity with kornia itself
cannot be pinned here: it is un-vendored, not installed, and there is no network).  What CAN be fixed independently
of any restatement is the geometry: for a plane the unprojected points are coplanar, so the normal is known in closed
form, and so are its orientation, the sign of the y flip the reference applies right after the call (T8:354) and the
replicate-padding behaviour at the border (of the unprojected POINTS, as kornia's spatial_gradient pads them).  Checked for the stand-alone kernel (depth_to_normals) AND for the stencil
fused into the march epilogue (render_from_depth), which share one device function.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda:0")
H, W, F_, OFF = 96, 128, 700.0, 1610.0


def camera():
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = F_
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    return K


def plane_depth(a, b, c0):
    """Depth map d(u,v) of the plane Z = c0 + a X + b Y seen through the pinhole X = (u-cx)/f d, Y = (v-cy)/f d:
    d = c0 / (1 - a (u-cx)/f - b (v-cy)/f).  The kernel adds OFF to the depth first (T8:353), so store d - OFF."""
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    d = c0 / (1.0 - a * (u - W / 2.0) / F_ - b * (v - H / 2.0) / F_)
    return (d - OFF).astype(np.float32)


def both_paths(depth_hw):
    """unit normals (3,H,W) from the stand-alone kernel and from the fused march epilogue."""
    from geomconsistentfr_amd.block import render_from_depth
    from geomconsistentfr_amd.normals import depth_to_normals
    d = torch.from_numpy(depth_hw)[None, None].to(DEV)
    n1 = depth_to_normals(d, camera(), negate_y=True, z_offset=OFF)[0].cpu().numpy()
    with torch.no_grad():
        o = render_from_depth(d, torch.full((1, 3, H, W), 0.5, device=DEV), torch.tensor([[0.2, 0.3, 0.9]], device=DEV),
                              torch.tensor([0.5], device=DEV), camera(), OFF, torch.ones(1, H, W, device=DEV))
    n2 = o["surface_normals"][0].cpu().numpy()
    assert np.array_equal(n1, n2)                       # one device function, same bits
    return n1


@pytest.mark.parametrize("a,b", [(0.0, 0.0), (0.3, 0.0), (0.0, -0.25), (0.2, 0.15), (-0.4, 0.1)])
def test_plane_gives_its_closed_form_normal(a, b):
    """Plane Z = c0 + a X + b Y: cross(dP/du, dP/dv) is parallel to (-a, -b, 1) (u grows with X, v with Y, both
    derivatives positive), and the reference negates y afterwards -> (-a, +b, 1) / |.| at EVERY pixel, border
    included (a Sobel stencil of coplanar points gives coplanar differences, replicate padding or not)."""
    n = both_paths(plane_depth(a, b, 1700.0))
    expect = np.array([-a, b, 1.0]) / np.sqrt(a * a + b * b + 1.0)
    err = np.abs(n - expect[:, None, None])
    # tolerance: the depth reaches the stencil as f32(depth + 1610) (T8:353), i.e. quantised to 1.2e-4 at 1700, against
    # a per-pixel difference of d/f = 2.4 -> slope errors of a few 1e-5; orientation and signs are pinned to 4 digits
    assert err[:, 1:-1, 1:-1].max() <= 1e-4, err[:, 1:-1, 1:-1].max()
    # border: replicate padding repeats the edge POINT (kornia pads the unprojected xyz, not the depth map), so the
    # one-sided differences still lie in the plane and only their lengths change -- the direction must not.  Zero
    # padding, or replicating the depth under a shifted pixel coordinate, would tilt every border normal.
    assert err.max() <= 1e-4, err.max()
    assert np.abs(np.linalg.norm(n, axis=0) - 1.0).max() <= 1e-6
    assert (n[2] > 0).all()                              # facing the camera (towards -Z in the reference's frame: z > 0 here)


def test_y_flip_sign_and_orientation_on_a_sphere_cap():
    """In the reference's frame x = c - W/2 grows to the right, y = H/2 - r grows UPWARDS (T8:52-53) and z is the
    depth value itself, largest where the face is nearest the viewer (the light sits at z = +4013 u_z, T8:362, and
    the Lambert term at T8:366 needs n.l > 0 on lit skin) -- so the normal the block uses is the OUTWARD normal of
    the height field z = depth(x, y):  (-dz/dx, -dz/dy, 1) / |.|.  On a dome (depth largest at the centre) it must
    tilt AWAY from the centre: x component of the sign of (u - cx), and, after the reference's y negation (T8:354),
    y component positive ABOVE the centre, i.e. of the sign of -(v - cy)."""
    v, u = np.mgrid[0:H, 0:W].astype(np.float64)
    R = 400.0
    x, y = u - W / 2.0, v - H / 2.0
    depth = (np.sqrt(R * R - x * x - y * y) - R + 60.0).astype(np.float32)   # 60 at the centre, falling off outwards
    n = both_paths(depth)
    inner = (slice(None), slice(4, H - 4), slice(4, W - 4))
    nx, ny, nz = n[inner]
    xs, ys = x[4:-4, 4:-4], y[4:-4, 4:-4]
    assert (nz > 0.5).all()
    assert (np.sign(nx[np.abs(xs) > 3]) == np.sign(xs[np.abs(xs) > 3])).all()
    assert (np.sign(ny[np.abs(ys) > 3]) == -np.sign(ys[np.abs(ys) > 3])).all()
    # and without the flip the sign is the opposite (negate_y is the T8:354 line, not part of kornia's function)
    from geomconsistentfr_amd.normals import depth_to_normals
    raw = depth_to_normals(torch.from_numpy(depth)[None, None].to(DEV), camera(), negate_y=False, z_offset=OFF)[0].cpu().numpy()
    assert np.array_equal(raw[1], -n[1]) and np.array_equal(raw[0], n[0]) and np.array_equal(raw[2], n[2])
