"""TEST INFRASTRUCTURE -- ctypes/numpy front end of the C oracle (oracle/gcfr_oracle.c).

Importable only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
All arrays are numpy, C-contiguous; nothing here touches torch or the GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libgcfr_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    src = os.path.join(_HERE, "gcfr_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "-B"])
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        c_f, c_d, c_i = ctypes.c_float, ctypes.c_double, ctypes.c_int
        p = ctypes.c_void_p
        L.gcfr_oracle_sample_table.argtypes = [c_d, c_d, c_i, p]
        L.gcfr_oracle_sample_table.restype = None
        L.gcfr_oracle_light_prep.argtypes = [p, c_i, c_i, c_f, c_f, p, p]
        L.gcfr_oracle_light_prep.restype = None
        L.gcfr_oracle_shadow_min_distance.argtypes = [p, p, c_i, p, c_i, c_i, c_i, c_i, c_i, p,
                                                      c_f, c_f, c_f, c_f, c_f, p, p]
        L.gcfr_oracle_shadow_min_distance.restype = None
        L.gcfr_oracle_shade.argtypes = [p, p, p, p, p, p, c_i, c_i, c_i, c_i, c_f, p, p, p, p, p]
        L.gcfr_oracle_shade.restype = None
        L.gcfr_oracle_num_threads.restype = c_i
        _lib = L
    return _lib


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


def num_threads() -> int:
    return int(lib().gcfr_oracle_num_threads())


def sample_table(t0=0.025, dt=0.005, n=160):
    out = np.empty(n, dtype=np.float64)
    lib().gcfr_oracle_sample_table(t0, dt, n, _ptr(out))
    return out


def light_prep(light_raw, clamp_z_min=0.0, light_distance=4013.0):
    """light_raw (B,3) -> (unit (B,3), light_pt (B,3)).  clamp_z_min=None: no clamp (inference)."""
    lr = _c(light_raw, np.float32).reshape(-1, 3)
    B = lr.shape[0]
    unit = np.empty((B, 3), np.float32)
    pt = np.empty((B, 3), np.float32)
    lib().gcfr_oracle_light_prep(_ptr(lr), B, int(clamp_z_min is not None),
                                 float(clamp_z_min or 0.0), float(light_distance), _ptr(unit), _ptr(pt))
    return unit, pt


def shadow_min_distance(depth, mask_u8, light_pt, t_table, bonus=0.0, bonus_box=None):
    """depth (B,H,W) f32, mask_u8 (B|1,H,W), light_pt (B,L,3) -> (min_dist, argmin) (B,L,H,W)."""
    depth = _c(depth, np.float32)
    B, H, W = depth.shape
    mask_u8 = _c(mask_u8, np.uint8).reshape(-1, H, W)
    light_pt = _c(light_pt, np.float32).reshape(B, -1, 3)
    L = light_pt.shape[1]
    t_table = _c(t_table, np.float64)
    md = np.empty((B, L, H, W), np.float32)
    am = np.empty((B, L, H, W), np.int32)
    box = bonus_box if bonus_box is not None else (0.0, -1.0, 0.0, -1.0)
    lib().gcfr_oracle_shadow_min_distance(_ptr(depth), _ptr(mask_u8), mask_u8.shape[0], _ptr(light_pt),
                                          B, L, H, W, len(t_table), _ptr(t_table), float(bonus),
                                          float(box[0]), float(box[1]), float(box[2]), float(box[3]),
                                          _ptr(md), _ptr(am))
    return md, am


def shade(normals, depth, albedo, light_pt, ambient, min_dist, intensity=0.5):
    """-> dict(shadow_w f32, full_shading f64, final_shading f64, rendered f32, normals f64)."""
    depth = _c(depth, np.float32)
    B, H, W = depth.shape
    normals = _c(normals, np.float64).reshape(B, 3, H, W)
    albedo = _c(albedo, np.float32).reshape(B, 3, H, W)
    light_pt = _c(light_pt, np.float32).reshape(B, -1, 3)
    L = light_pt.shape[1]
    ambient = _c(ambient, np.float32).reshape(B, L)
    min_dist = _c(min_dist, np.float32).reshape(B, L, H, W)
    w = np.empty((B, L, H, W), np.float32)
    full = np.empty((B, L, H, W), np.float64)
    fin = np.empty((B, L, H, W), np.float64)
    ren = np.empty((B, L, 3, H, W), np.float32)
    nout = np.empty((B, 3, H, W), np.float64)
    lib().gcfr_oracle_shade(_ptr(normals), _ptr(depth), _ptr(albedo), _ptr(light_pt), _ptr(ambient),
                            _ptr(min_dist), B, L, H, W, float(intensity), _ptr(w), _ptr(full), _ptr(fin),
                            _ptr(ren), _ptr(nout))
    return dict(shadow_w=w, full_shading=full, final_shading=fin, rendered=ren, normals=nout)
