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
ABI_VERSION = 6      # include/gcfr.h GCFR_ABI_VERSION this binding was written against
_ERRORS = {-1: "GCFR_ERR_INVALID_ARGUMENT", -2: "GCFR_ERR_LAUNCH"}

_p, _i, _f, _d = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float, ctypes.c_double

N_COUNTERS = 28      # GCFR_N_COUNTERS
COUNTER_NAMES = ("tiles", "groups_nominal", "groups_visited", "bound_tests", "bodies", "lane_samples", "early_exits",
                 "tie_remarches", "samples_in_range", "bounds_given_up", "visits_after_last_body", "visits_before_first_body",
                 "trail_enter", "trail_skips", "trail_leave", "rough_samples", "wave_samples", "wave_samples_taken", "lane_takes",
                 # the audit build (-DGCFR_AUDIT, csrc/gcfr_march.hpp): claims checked / contradicted; audit_max_use is a maximum (1/1000 of Kerr)
                 "audit_bound_checks", "audit_bound_violations", "audit_term_checks", "audit_term_violations",
                 "audit_masked_checks", "audit_masked_violations", "audit_safe_violations", "audit_max_use")


class Options(ctypes.Structure):
    """include/gcfr.h `gcfr_options`: per-call knobs and hooks of the forward entry points (none changes a result bit
    except `pixels`, see the header).  Build one with `options(...)`; pass it as `options=` to the block functions /
    RenderFwdPlan."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("tile_w", _i), ("group", _i), ("ksplit", _i),
                ("depth_bound_skip", _i), ("lds_stage", _i),
                ("event_start", _p), ("event_stop", _p), ("counters", _p), ("pixels", _i), ("phase", _i)]


def options(tile_w=0, group=0, ksplit=-1, depth_bound_skip=-1, event_start=None,
            event_stop=None, counters=None, lds_stage=-1, pixels=0, phase=0) -> Options:
    o = Options()
    load().gcfr_options_default(ctypes.byref(o))
    o.tile_w, o.group, o.ksplit, o.depth_bound_skip = tile_w, group, ksplit, depth_bound_skip
    o.lds_stage = lds_stage
    o.event_start, o.event_stop, o.counters = event_start, event_stop, counters
    o.pixels = pixels
    o.phase = phase
    return o


def with_phase(o, phase: int) -> Options:
    """a copy of `o` (or the defaults) with `phase` set: 1 = the prepass only, 2 = the march only (include/gcfr.h)"""
    n = Options()
    if o is None:
        load().gcfr_options_default(ctypes.byref(n))
    else:
        ctypes.memmove(ctypes.byref(n), ctypes.byref(o), ctypes.sizeof(Options))
    n.phase = phase
    return n


def with_pixels(o, pixels: int) -> Options:
    """a copy of `o` (or the defaults) with `pixels` set -- RenderParams(pixels="mask") reaches the library this way"""
    n = Options()
    if o is None:
        load().gcfr_options_default(ctypes.byref(n))
    else:
        ctypes.memmove(ctypes.byref(n), ctypes.byref(o), ctypes.sizeof(Options))
    n.pixels = pixels
    return n


def opt_ref(o):
    """ctypes argument for a `const gcfr_options *` parameter (None = library defaults)."""
    return ctypes.byref(o) if o is not None else None


_SIGNATURES = {
    "gcfr_version": (ctypes.c_char_p, []),
    "gcfr_abi_version": (_i, []),
    "gcfr_sample_table": (_i, [_d, _d, _i, _p]),
    "gcfr_light_prep": (_i, [_p, _i, _i, _f, _f, _p, _p, _p]),
    "gcfr_shadow_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "gcfr_options_default": (None, [_p]),
    "gcfr_render_from_depth_fwd": (_i, [_p, _i, _f, _f, _p, _p, _i, _d, _d, _d, _d, _f, _i, _p, _p, _i, _i, _i, _i, _i, _p,
                                        _f, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p, _p]),
    "gcfr_normals_fwd": (_i, [_p, _i, _i, _i, _d, _d, _d, _d, _f, _i, _p, _p]),
    "gcfr_normals_bwd": (_i, [_p, _p, _i, _i, _i, _d, _d, _d, _d, _f, _i, _p, _p]),
    "gcfr_render_fwd": (_i, [_p, _i, _f, _f, _p, _p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _p, _f, _p, _f,
                             _p, _p, _p, _p, _p, _p, _p, _p, _p, ctypes.c_size_t, _p, _p]),
    "gcfr_shadow_fwd": (_i, [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p, _p, ctypes.c_size_t, _p, _p]),
    "gcfr_shade_fwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p]),
    "gcfr_shadow_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p]),
    "gcfr_shade_bwd": (_i, [_p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p]),
    "gcfr_render_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _d, _d, _d, _d, _f, _i, _f, _p, _p, _p, _p, _p,
                             _p, _p, _p, _p, _p]),
    "gcfr_light_prep_bwd": (_i, [_p, _i, _i, _f, _f, _p, _p, _p, _p]),
    "gcfr_inference_images_u8": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p]),
    "gcfr_fix_border_u8": (_i, [_p, _p, _i, _i, _i, _i, _p, _p]),
    "gcfr_assemble_batch_u8": (_i, [_p, _p, _p, _p, _i, _i, _i, _p, _p, _p, _p, _p]),
    "gcfr_masked_metrics_workspace_bytes": (ctypes.c_size_t, [_i, _i, _i]),
    "gcfr_masked_metrics_u8": (_i, [_p, _p, _p, _i, _i, _i, _i, _p, _p, _p, ctypes.c_size_t, _p]),
    "gcfr_copy_probe": (_i, [_p, _p, ctypes.c_size_t, _p]),
}


class GcfrError(RuntimeError):
    pass


def lib_path() -> str:
    return _build.LIB_PATH


def load():
    """Load and type the library.  Without an override the in-tree library is (re)built first whenever it is
    missing or older than a source under csrc/ or include/gcfr.h and hipcc is available; a stale library on a
    host without hipcc (the GPU box ships the prebuilt .so) is loaded as it is.  `GCFR_HIP_LIB` selects another
    build of the same ABI (tools/ab.sh); a missing override is an error, never a silent fall-back."""
    global _LIB
    if _LIB is not None:
        return _LIB
    override = os.environ.get("GCFR_HIP_LIB")
    if override:
        if not os.path.exists(override):
            raise GcfrError("GCFR_HIP_LIB=%s does not exist" % override)
        path = override
    else:
        path = _build.LIB_PATH
        if _build.needs_build():
            try:
                _build.build()
            except Exception as e:  # no hipcc on this host
                if not os.path.exists(path):
                    raise GcfrError("libgcfr_hip.so is not built (%s) and there is no CPU fallback: run "
                                    "`python -c 'import __graft_entry__ as g; g.build()'`" % e)
                import warnings
                warnings.warn("libgcfr_hip.so is older than its sources and could not be rebuilt (%s): loading the stale "
                              "library (its ABI revision is checked below)" % e, RuntimeWarning)
    L = ctypes.CDLL(path)
    # ABI revision first: a stale library exports every symbol by name, and would be called with shifted arguments
    try:
        L.gcfr_abi_version.restype = _i
        L.gcfr_abi_version.argtypes = []
        have = int(L.gcfr_abi_version())
    except AttributeError:
        have = None
    if have != ABI_VERSION:
        raise GcfrError("%s implements ABI revision %s, this binding needs %d (include/gcfr.h GCFR_ABI_VERSION): rebuild "
                        "with `python -c 'import __graft_entry__ as g; g.build()'`" % (path, have, ABI_VERSION))
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
