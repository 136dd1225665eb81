"""ctypes binding of libgcfr_hip.so (C ABI: include/gcfr.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  torch is
imported first on purpose -- it loads its bundled libamdhip64.so (SONAME libamdhip64.so.7), and the
dynamic loader then binds libgcfr_hip.so's DT_NEEDED libamdhip64.so.7 to that same, already loaded
runtime, so device pointers and streams are shared between torch and the kernels.
"""
import ctypes
import os

import torch  # noqa: F401  (must be loaded before libgcfr_hip.so, see above)

from . import build as _build

_LIB = None

GCFR_OK = 0
_ERRORS = {-1: "GCFR_ERR_INVALID_ARGUMENT", -2: "GCFR_ERR_LAUNCH"}

_p, _i, _f, _d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_double

_SIGNATURES = {
    "gcfr_version": (ctypes.c_char_p, []),
    "gcfr_sample_table": (_i, [_d, _d, _i, _p]),
    "gcfr_light_prep": (_i, [_p, _i, _i, _f, _f, _p, _p, _p]),
    "gcfr_shadow_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "gcfr_tune": (_i, [_i, _i]),
    "gcfr_profile_events": (_i, [_p, _p]),
    "gcfr_render_from_depth_fwd": (_i, [_p, _i, _f, _f, _p, _p, _i, _d, _d, _d, _d, _f, _i, _p, _p, _i, _i, _i, _i, _i, _p,
                                        _f, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "gcfr_normals_fwd": (_i, [_p, _i, _i, _i, _d, _d, _d, _d, _f, _i, _p, _p]),
    "gcfr_normals_bwd": (_i, [_p, _p, _i, _i, _i, _d, _d, _d, _d, _f, _i, _p, _p]),
    "gcfr_render_fwd": (_i, [_p, _i, _f, _f, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p, _f,
                             _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "gcfr_shadow_fwd": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p, _p, ctypes.c_size_t, _p]),
    "gcfr_shade_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "gcfr_shadow_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "gcfr_shade_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gcfr_render_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _d, _d, _d, _d, _f, _i, _f, _p, _p, _p, _p, _p,
                             _p, _p, _p, _p, _p]),
    "gcfr_light_prep_bwd": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p]),
}


class GcfrError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load (building first if the sources are newer and hipcc is available) and type the library."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.environ.get("GCFR_HIP_LIB") or _build.LIB_PATH  # override: A/B builds of the same ABI (tools/ab.sh)
    if not os.path.exists(path):
        try:
            _build.build()
        except Exception as e:  # no hipcc on this host
            raise GcfrError("libgcfr_hip.so is not built (%s) and there is no CPU fallback: run "
                            "`python -c 'import __graft_entry__ as g; g.build()'`" % e)
    L = ctypes.CDLL(path)
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(L, name)  # AttributeError here = the library does not export the declared ABI
        fn.restype = res
        fn.argtypes = args
    _LIB = L
    return L


def exported_symbols():
    return sorted(_SIGNATURES)


def check(status: int, what: str):
    if status != GCFR_OK:
        raise GcfrError("%s failed: %s (%d)" % (what, _ERRORS.get(status, "unknown"), status))
