"""Synthetic / fixture-derived input scenes shared by bench.py, the GPU parity tests and the tools.

One module so that the parity tests' inputs cannot drift with the bench: tests import `scenes`, not `bench`, and
tests/test_scenes_host.py pins what these generators return for fixed seeds (tests/golden/scenes_pin.npz).
Not product code: the render block never sees where its inputs came from.
"""
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H = W = 256


LIGHTS18 = np.array([[.7518, 0, .6594], [.6893, .3991, .6047], [.5145, 0, .8575], [-.5843, 0, .8115],
                     [-.7574, 0, .6529], [-.7076, .3892, .5897], [-.5151, .4722, .7154], [.4478, .4925, .7463],
                     [0, .7071, .7071], [-.8138, -.3420, .4698], [.8138, -.3420, .4698],      # 11 from S1:519-562
                     [.3, .3, .9], [-.3, .3, .9], [.2, -.5, .84], [-.2, -.5, .84], [.9, .1, .42], [-.9, .1, .42],
                     [0, .2, .98]], np.float32)                                                # 7 synthetic


def synth_faces_sized(B, seed0, size, n_lights, mask_kind="ellipse", light_seed0=None):
    """Config-5 style inputs: `size` x `size` faces (surface scaled), `n_lights` lights per face."""
    r, c = np.mgrid[0:size, 0:size]
    s = size / 256.0
    x, y = (c - size / 2.0) / s, (r - size / 2.0) / s
    depth, mask, albedo, normals = [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        ax, ay, nose = 85 + 10 * rng.random(), 105 + 10 * rng.random(), 30 + 10 * rng.random()
        d = s * (80 * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0))
                 + nose * np.exp(-(x ** 2 / 288 + (y - 12) ** 2 / 648)) + 3 * np.sin(x / 7) * np.cos(y / 9))
        depth.append(d.astype(np.float32))
        m = (((x / (ax - 8)) ** 2 + (y / (ay - 8)) ** 2) < 1) if mask_kind == "ellipse" else (
            np.ones_like(x, bool) if mask_kind == "ones" else np.zeros_like(x, bool))
        mask.append(m.astype(np.uint8))
        albedo.append((0.15 + 0.7 * rng.random((3, size, size))).astype(np.float32))
        gy, gx = np.gradient(d)
        n = np.stack([-gx, gy, np.ones_like(d)])
        normals.append((n / np.linalg.norm(n, axis=0)).astype(np.float32))
    ls0 = seed0 if light_seed0 is None else light_seed0
    light = np.stack([np.roll(LIGHTS18, ls0 + i, axis=0)[:n_lights] for i in range(B)])
    amb = np.full((B, n_lights), 0.5, np.float32)
    return np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), light, amb


def synth_faces(B, seed0, light_seed0=None):
    """Deterministic synthetic faces: jittered ellipsoid + nose + ripple.
    Face i takes light (light_seed0 + i) mod 11 of the reference's eleven shipped directions (light_seed0 = seed0 unless given)."""
    r, c = np.mgrid[0:H, 0:W]
    x, y = c - 128.0, r - 128.0
    lights11 = np.array([[.7518, 0, .6594], [.6893, .3991, .6047], [.5145, 0, .8575], [-.5843, 0, .8115],
                         [-.7574, 0, .6529], [-.7076, .3892, .5897], [-.5151, .4722, .7154], [.4478, .4925, .7463],
                         [0, .7071, .7071], [-.8138, -.3420, .4698], [.8138, -.3420, .4698]], np.float32)
    depth, mask, albedo, normals, light, amb = [], [], [], [], [], []
    for i in range(B):
        rng = np.random.default_rng(seed0 + i)
        ax, ay, nose = 85 + 10 * rng.random(), 105 + 10 * rng.random(), 30 + 10 * rng.random()
        d = 80 * np.sqrt(np.maximum(1 - (x / ax) ** 2 - (y / ay) ** 2, 0)) \
            + nose * np.exp(-(x ** 2 / 288 + (y - 12) ** 2 / 648)) + 3 * np.sin(c / 7) * np.cos(r / 9)
        depth.append(d.astype(np.float32))
        mask.append((((x / (ax - 8)) ** 2 + (y / (ay - 8)) ** 2) < 1).astype(np.uint8))
        albedo.append((0.15 + 0.7 * rng.random((3, H, W))).astype(np.float32))
        gy, gx = np.gradient(d)
        n = np.stack([-gx, gy, np.ones_like(d)])
        normals.append((n / np.linalg.norm(n, axis=0)).astype(np.float32))
        light.append(lights11[((seed0 if light_seed0 is None else light_seed0) + i) % 11])
        amb.append(np.float32(0.5))
    return (np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), np.stack(light),
            np.asarray(amb, np.float32))


def ffhq_faces(B, first):
    """`--data ffhq`: the three checkpoint-derived FFHQ depth maps and skin masks of the golden fixtures
    (tests/golden/inputs.npz: sample_test_images_FFHQ/{00295,00110,00508}.png through the reference's lighting-transfer
    network + shipped checkpoint, oracle/make_golden.py) tiled to B faces: face g = first + i uses fixture g mod 3,
    mirrored left-right on every other pass through the three (a mirrored face is a face); the fixture albedo;
    normals by finite differences of the depth (the render block takes normals as an input)."""
    g = os.path.join(ROOT, "tests", "golden")
    inp = np.load(os.path.join(g, "inputs.npz"))
    alb0 = np.load(os.path.join(g, "albedo.npz"))["albedo"]
    depths, masks = inp["depths"][1:4], inp["masks"][2:5]
    lights11 = LIGHTS18[:11]
    depth, mask, albedo, normals, light, amb = [], [], [], [], [], []
    for i in range(B):
        gi = first + i
        d, m, al = depths[gi % 3], masks[gi % 3], alb0
        if (gi // 3) % 2 == 1:
            d, m, al = d[:, ::-1], m[:, ::-1], al[:, :, ::-1]
        d = np.ascontiguousarray(d, np.float32)
        depth.append(d)
        mask.append(np.ascontiguousarray(m, np.uint8))
        albedo.append(np.ascontiguousarray(al, np.float32))
        gy, gx = np.gradient(d.astype(np.float64))
        n = np.stack([-gx, gy, np.ones_like(gx)])
        normals.append((n / np.linalg.norm(n, axis=0)).astype(np.float32))
        light.append(lights11[i % 11])
        amb.append(np.float32(0.5))
    return (np.stack(depth), np.stack(mask), np.stack(albedo), np.stack(normals), np.stack(light),
            np.asarray(amb, np.float32))
