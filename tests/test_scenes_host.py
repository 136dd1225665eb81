"""CPU: tools/scenes.py -- the input scenes bench.py, the GPU parity tests and the tools share -- returns what it returned
when tests/golden/scenes_pin.npz was written (round 6), for fixed seeds.  An edit to a generator therefore fails HERE instead
of silently changing what the parity tests compare and what the bench times.  Depth / normals are compared with a tolerance
(libm's sqrt / exp / sin differ in the last bit between hosts), everything drawn from numpy's Generator and every count exactly.
"""
import os

import numpy as np

import scenes

PIN = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scenes_pin.npz"))


def test_synth_faces_are_pinned():
    d, m, a, n, l, amb = scenes.synth_faces(3, 11)
    assert d.dtype == np.float32 and m.dtype == np.uint8 and d.shape == (3, 256, 256)
    np.testing.assert_allclose(d[:, ::16, ::16], PIN["sf_depth"], atol=1e-4)
    np.testing.assert_array_equal(m.reshape(3, -1).sum(1), PIN["sf_mask_sum"])
    np.testing.assert_array_equal(a[:, :, ::32, ::32], PIN["sf_albedo"])
    np.testing.assert_allclose(n[:, :, ::32, ::32], PIN["sf_normals"], atol=1e-5)
    np.testing.assert_array_equal(l, PIN["sf_light"])
    np.testing.assert_array_equal(amb, PIN["sf_amb"])


def test_synth_faces_sized_are_pinned():
    d, m, a, n, l, amb = scenes.synth_faces_sized(1, 3, 512, 18)
    assert l.shape == (1, 18, 3) and amb.shape == (1, 18)
    np.testing.assert_allclose(d[:, ::32, ::32], PIN["sz_depth"], atol=1e-4)
    np.testing.assert_array_equal(m.reshape(1, -1).sum(1), PIN["sz_mask_sum"])
    np.testing.assert_array_equal(a[:, :, ::64, ::64], PIN["sz_albedo"])
    np.testing.assert_array_equal(l, PIN["sz_light"])
    np.testing.assert_array_equal(amb, PIN["sz_amb"])
    d, m, a, n, l, amb = scenes.synth_faces_sized(2, 5, 128, 4, mask_kind="ones", light_seed0=2)
    np.testing.assert_allclose(d[:, ::8, ::8], PIN["so_depth"], atol=1e-4)
    np.testing.assert_array_equal(m.reshape(2, -1).sum(1), PIN["so_mask_sum"])
    np.testing.assert_array_equal(l, PIN["so_light"])


def test_ffhq_faces_are_pinned_and_bench_uses_this_module():
    d, m, a, n, l, amb = scenes.ffhq_faces(5, 2)
    np.testing.assert_array_equal(d[:, ::16, ::16], PIN["ff_depth"])          # fixture data, mirrored / tiled: exact
    np.testing.assert_array_equal(m.reshape(5, -1).sum(1), PIN["ff_mask_sum"])
    np.testing.assert_array_equal(l, PIN["ff_light"])
    import bench
    assert bench.synth_faces is scenes.synth_faces and bench.synth_faces_sized is scenes.synth_faces_sized
    assert bench.ffhq_faces is scenes.ffhq_faces and bench.LIGHTS18 is scenes.LIGHTS18
