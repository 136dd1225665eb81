"""Loader for tests/golden/*.npz (generated from the reference by oracle/make_golden.py).

Every case is flattened to per-(face, light) records with uniform keys so that oracle tests and
GPU parity tests iterate over the same list:
  variant, name, depth (H,W) f32, mask (H,W) u8, albedo (3,H,W) f32, light (3,), ambient (scalar),
  params (dict: n_samples,t0,dt,intensity,clamp_light_z_min,normal_z_offset,focal,bonus,bonus_box),
  expect: dict of reference outputs for that face.  `minimum_distance` / `argmin` are what the reference's own
  torch.min returned at T8:514 (S1:492, SLT:499), i.e. BEFORE the inference forms' +5 bonus; where the minimum is a
  masked sample (value 1e6) `argmin` is torch.min's first masked index (the product reports -1 there).
"""
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
H = W = 256

T8_PARAMS = dict(n_samples=160, t0=0.025, dt=0.005, intensity=0.5, clamp_light_z_min=0.0,
                 normal_z_offset=1610.0, focal=1570.0, bonus=0.0, bonus_box=None, light_distance=4013.0)
S1_PARAMS = dict(T8_PARAMS, clamp_light_z_min=None, bonus=5.0,
                 bonus_box=(-(W / 2.0), W - W / 2.0 - 1, 1 - H / 2.0, H / 2.0))          # S1:495
SLT_PARAMS = dict(T8_PARAMS, n_samples=159, t0=0.03, intensity=0.41, clamp_light_z_min=None,
                  normal_z_offset=1410.0, focal=700.0, bonus=5.0,
                  bonus_box=(-4.0 * W, 4.0 * W, 4.0 * (1 - H), 4.0 * H))                 # SLT:503


def _inputs():
    inp = np.load(os.path.join(GOLDEN, "inputs.npz"))
    alb = np.load(os.path.join(GOLDEN, "albedo.npz"))["albedo"]
    return inp["depths"], inp["masks"], alb


SMOOTH_T8 = ("t8_a", "t8_b", "t8_c", "t8_d", "t8_e")
# oracle/make_golden_rough.py: untrained-network / noise-40 / noise-400 depth, sparse masks, lights on the nine-way
# branch's boundaries (t8_f exactly on them, t8_h one ulp outside), (1,0,0) / (0,0,1) / z<0 (t8_g).  t8_f stores the RGB
# of face 0 only (`rendered_images_face0`) and the gradients of the `full` loss; t8_h the march's values/indices only.
ROUGH_T8 = ("t8_f", "t8_g", "t8_h")


def t8_batches():
    """Yield (name, dict) per T8 batch of 3 faces (training form): the smooth batches, then the rough ones."""
    depths_s, masks_s, alb = _inputs()
    rough = np.load(os.path.join(GOLDEN, "inputs_rough.npz"))
    for name in SMOOTH_T8 + ROUGH_T8:
        depths, masks = (depths_s, masks_s) if name in SMOOTH_T8 else (rough["depths"], rough["masks"])
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        di, mi = z["depth_idx"], z["mask_idx"]
        yield name, dict(
            depth=depths[di], mask=masks[mi], albedo=np.stack([np.roll(alb, b, axis=0) for b in range(3)]),
            light=z["light4"][:, 1:4].copy(), ambient=z["light4"][:, 0].copy(), params=T8_PARAMS,
            expect={k: z[k] for k in z.files if k not in ("depth_idx", "mask_idx", "light4", "albedo_roll")})


def single_cases():
    """Yield (name, dict) per B=1 inference case (S1 and SLT forms)."""
    depths, masks, alb = _inputs()
    for name in ("s1_a", "s1_b", "s1_c", "s1_d", "s1_e", "slt_a", "slt_b"):
        z = np.load(os.path.join(GOLDEN, name + ".npz"))
        s1 = name.startswith("s1")
        amb = (z["raw4"][0] - np.float32(0.1)) if s1 else z["target_ambient"]          # S1:342 / SLT:348
        yield name, dict(
            depth=depths[int(z["depth_idx"])][None], mask=masks[int(z["mask_idx"])][None], albedo=alb[None],
            light=z["target_light"][None].copy(), ambient=np.asarray([amb], np.float32),
            params=S1_PARAMS if s1 else SLT_PARAMS,
            expect={k: z[k][None] for k in ("shadow_mask_weights", "full_shading", "rendered_images",
                                            "unit_light_direction", "minimum_distance", "argmin") if k in z.files})


def all_cases():
    yield from t8_batches()
    yield from single_cases()
