"""CPU: callers / data formats either side of the block against independent
restatements of the reference lines they follow (scipy.ndimage / explicit loops)."""
import numpy as np
import scipy.io
import scipy.ndimage as ndi

from geomconsistentfr_amd import postprocess as pp


def test_to_uint8_rounds_half_to_even_and_saturates():
    np.testing.assert_array_equal(pp.to_uint8(np.array([-3.0, 0.5, 1.5, 2.5, 254.5, 255.5, 300.0])),
                                  [0, 0, 2, 2, 254, 255, 255])


def test_composite_into_input_matches_a_per_pixel_restatement():
    rng = np.random.default_rng(0)
    H, W = 12, 9
    inp = rng.random((H, W, 3))
    ren = rng.random((3, H, W))
    mask = rng.choice([0.0, 64 / 255.0, 128 / 255.0, 1.0], size=(H, W))
    got = pp.composite_into_input(inp, ren, mask)
    for r in range(H):
        for c in range(W):
            for ch in range(3):
                # S1:616-619: 255.0*rendered is an f32 product (rendered_images is an f32 array), widened by the f64 mask
                exp = float(np.float32(255.0) * np.float32(ren[ch, r, c])) * mask[r, c] if mask[r, c] > 0 else inp[r, c, ch] * 255.0
                assert got[r, c, ch] == exp


def test_diagnostic_images_follow_s8():
    rng = np.random.default_rng(1)
    B, H, W = 2, 8, 8
    depth = rng.standard_normal((B, 1, H, W)) * 30
    mask = (rng.random((H, W)) > 0.4).astype(np.float64)
    out = pp.diagnostic_images(rng.random((H, W, 3)), rng.random((3, H, W)), depth, 1, rng.random((H, W)),
                               rng.random((3, H, W)), rng.random((H, W)), rng.standard_normal((3, H, W)), mask)
    assert set(out) == {"rendered_image", "shadow_mask", "albedo", "depth", "shading", "surface_normals"}
    d = -depth.astype(np.float32)                                             # the forward's outputs are f32 arrays
    d = (d - d.min()) / (d.max() - d.min())                                   # S8:589-590: over the batch
    np.testing.assert_array_equal(out["depth"], (np.float32(255.0) * d[1, 0]).astype(np.float64) * mask)
    assert out["surface_normals"].shape == (H, W, 3) and out["shadow_mask"].shape == (H, W)
    assert np.all(out["albedo"][mask == 0] == 0)


def test_fix_border_artifacts_matches_scipy_restatement():
    rng = np.random.default_rng(2)
    H, W = 40, 36
    img = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
    mask = np.zeros((H, W), np.uint8)
    mask[8:30, 6:28] = rng.choice([64, 128, 255], size=(22, 22)).astype(np.uint8)
    got = pp.fix_border_artifacts(img, mask)
    m = np.floor(mask / 255.0 + 0.5)                                           # MATLAB uint8 division
    conv = ndi.correlate(m, np.ones((7, 7)), mode="constant", cval=0.0)
    border = (conv < 30) & (conv > 0)
    exp = img.copy()
    for c in range(3):
        med = ndi.median_filter(img[..., c], size=3, mode="constant", cval=0)
        exp[..., c][border] = med[border]
    np.testing.assert_array_equal(got, exp)
    assert border.any() and not border.all()
    np.testing.assert_array_equal(got[~border], img[~border])


def test_masked_mse():
    rng = np.random.default_rng(3)
    a = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    b = rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)
    m = rng.choice([0, 255], size=(16, 16)).astype(np.uint8)
    exp = (((a / 255.0 - b / 255.0) ** 2) * (m[..., None] / 255.0) ** 2).sum() / (3 * (m / 255.0).sum())
    assert abs(pp.masked_mse(a, b, m) - exp) < 1e-15
    assert pp.masked_mse(a, a, m) == 0.0


def test_masked_dssim_against_scipy_gaussian():
    rng = np.random.default_rng(4)
    a = rng.integers(0, 256, (32, 32, 3), dtype=np.uint8)
    b = np.clip(a.astype(int) + rng.integers(-20, 21, a.shape), 0, 255).astype(np.uint8)
    m = np.zeros((32, 32), np.uint8)
    m[6:26, 5:27] = 255
    assert abs(pp.masked_dssim(a, a, m)) < 1e-12
    A, R = a / 255.0, b / 255.0
    g = lambda x: ndi.gaussian_filter(x, 1.5, mode="nearest", truncate=3.4)    # radius 5 on all three axes
    C1, C2 = 1e-4, 9e-4
    mx, my = g(A), g(R)
    sx, sy, sxy = g(A * A) - mx * mx, g(R * R) - my * my, g(A * R) - mx * my
    smap = ((2 * mx * my + C1) * (2 * sxy + C2)) / ((mx * mx + my * my + C1) * (sx + sy + C2))
    m3 = np.repeat((m / 255.0)[..., None], 3, 2)
    exp = (1 - (smap * m3).sum() / m3.sum()) / 2
    assert abs(pp.masked_dssim(a, b, m) - exp) < 1e-12
    assert 0 < pp.masked_dssim(a, b, m) < 0.5


def test_on_disk_formats(tmp_path):
    rng = np.random.default_rng(5)
    depth = rng.standard_normal((256, 256)) * 40
    scipy.io.savemat(tmp_path / "00001_depth.mat", {"depth_img": depth})                 # T8:545
    scipy.io.savemat(tmp_path / "00001.jpg.mat", {"lighting_direction": np.array([[0.1, 0.2, 0.97]])})   # T8:549
    d = pp.load_depth_mat(str(tmp_path / "00001_depth.mat"))
    assert d.shape == (256, 256, 1) and d.dtype == np.float64
    np.testing.assert_array_equal(d[..., 0], depth)
    np.testing.assert_allclose(pp.load_lighting_mat(str(tmp_path / "00001.jpg.mat")), [0.5, 0.1, 0.2, 0.97])
    face = rng.integers(0, 256, (256, 256), dtype=np.uint8)
    dm = rng.choice([0, 64, 128, 255], size=(256, 256)).astype(np.uint8)
    got = pp.fill_nose_and_mouth_mask(face, dm)
    tmp = np.maximum(face.astype(np.float64), dm.astype(np.float64))                       # T8:553-555
    exp = np.where(tmp > 128, 255.0, 0.0)
    np.testing.assert_array_equal(got[..., 0], exp)
    assert set(np.unique(got)) <= {0.0, 255.0}
