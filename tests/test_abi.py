"""CPU: the C-ABI library loads and exports every symbol include/gcfr.h declares (no compute calls)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "gcfr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gcfr_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    syms = declared_symbols()
    assert len(syms) >= 5
    for s in syms:
        assert hasattr(L, s), "libgcfr_hip.so does not export %s" % s
    assert set(_lib.exported_symbols()) == set(syms), "python binding and header disagree"
    assert b"gfx950" in L.gcfr_version()


def test_sample_table_host_helper_matches_numpy():
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    for t0, dt, n, stop in [(0.025, 0.005, 160, 0.825), (0.03, 0.005, 159, 0.825), (0.025, 0.0025, 320, 0.825)]:
        out = np.empty(n)
        assert L.gcfr_sample_table(t0, dt, n, out.ctypes.data) == 0
        np.testing.assert_array_equal(out, np.arange(t0, stop, dt))
    assert L.gcfr_sample_table(0.0, 0.1, 0, None) == -1


def test_invalid_arguments_are_rejected_without_a_gpu():
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    assert L.gcfr_shadow_fwd(None, None, 1, None, 1, 1, 256, 256, 160, None, 0.0, None, None, None, None, 0, None, None) == -1
    # workspace: quad texels + 4 statistics chunks per 256x256 image (box 16 B + depth range 8 B) + depth-bounds records
    # (round 3: the records' per-image stride is a whole number of 1-KiB pieces -- 1090 -> 1152 records -- and one mask bitmap of
    #  H*W/8 bytes per image follows, both for the LDS-staged march; behind each image's records 4 x 1024 float4 of horizon
    #  tables for the trailing loop's termination test)
    assert L.gcfr_shadow_workspace_bytes(8, 256, 256) == (8 * 257 * 257 * 16 + 8 * 4 * 16 + 8 * (1152 + 4096) * 16 + 8 * 4 * 8 + 65 * 4 + 12
                                                          + 8 * 4 * 4 + 16 + 8 * 8192 + 16 + 8 * 4 * 16)     # ... + the masks' partial diagonal extents
    assert L.gcfr_light_prep(None, 1, 1, 0.0, 4013.0, None, None, None) == -1
    assert L.gcfr_shade_fwd(None, None, None, None, None, None, 1, 1, 8, 8, 0.5, None, None, None, None, None) == -1
    # the inference image side: null planes, a diagnostic output without its input, mask batch neither 1 nor B, in-place border fix
    assert L.gcfr_inference_images_u8(None, None, None, None, None, None, None, None, None, 1, 1, 1, 8, 8, None, None, None, None, None, None, 0, None) == -1
    assert L.gcfr_inference_images_u8(16, 16, None, None, None, None, None, None, 16, 1, 1, 1, 8, 8, 16, 16, None, None, None, None, 0, None) == -1
    assert L.gcfr_inference_images_u8(16, 16, None, None, None, None, None, None, 16, 2, 3, 1, 8, 8, 16, None, None, None, None, None, 0, None) == -1
    assert L.gcfr_inference_images_u8(16, 16, None, None, None, None, None, None, 16, 1, 1, 1, 8, 8, 16, None, None, None, None, None, 2, None) == -1
    assert L.gcfr_inference_images_u8(16, 16, None, None, None, None, None, None, 16, 1, 1, 0, 8, 8, 16, None, None, None, None, None, 0, None) == -1   # L = 0
    assert L.gcfr_fix_border_u8(16, 16, 1, 1, 8, 8, 16, None) == -1


def test_options_struct_defaults_and_layout():
    """gcfr_options: the python mirror has the header's layout; defaults are all 'auto'; a caller built against
    another layout (wrong struct_size) or an out-of-range knob is rejected before anything is launched."""
    import ctypes
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    o = _lib.Options()
    L.gcfr_options_default(ctypes.byref(o))
    assert o.struct_size == ctypes.sizeof(_lib.Options) == 56      # ABI 6 (64 at revisions 4 and 5: they are refused, below)
    assert (o.tile_w, o.group, o.ksplit, o.depth_bound_skip, o.lds_stage) == (0, 0, -1, -1, -1)
    assert o.pixels == 0                 # the one result-changing knob is off unless asked for
    assert not o.event_start and not o.event_stop and not o.counters
    # argument validation happens on the host before any launch, so it can be exercised without a GPU: dummy non-null
    # pointers, a valid shape, then a bad options struct
    dummy = ctypes.c_void_p(64)
    args = (dummy, dummy, 1, dummy, 1, 1, 256, 256, 160, dummy, 0.0, None, dummy, None, None, 0, None)
    bad = _lib.options(tile_w=24)
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(bad)) == -1
    bad = _lib.options()
    bad.struct_size = 8
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(bad)) == -1
    # a caller built against revision 4 / 5 of the struct (64 bytes, `schedule` / `tile_order` in front of `lds_stage`): its
    # struct_size is refused BEFORE any other field is read -- also a `phase` of 1 at the old offset is never interpreted
    # (round-5 advisor finding: `opt->phase` was read ahead of the size check)
    class OldOptions(ctypes.Structure):
        _fields_ = [("struct_size", ctypes.c_uint32)] + [(n, ctypes.c_int32) for n in ("tile_w", "group", "ksplit", "depth_bound_skip",
                    "schedule", "tile_order", "lds_stage")] + [(n, ctypes.c_void_p) for n in ("event_start", "event_stop", "counters")] + \
                   [("pixels", ctypes.c_int32), ("phase", ctypes.c_int32)]
    old = OldOptions(struct_size=64, ksplit=-1, depth_bound_skip=-1, schedule=-1, tile_order=-1, lds_stage=-1, phase=1)
    assert ctypes.sizeof(OldOptions) == 64
    ws_args = args[:14] + (dummy, 1 << 30, None)
    assert L.gcfr_shadow_fwd(*ws_args, ctypes.byref(old)) == -1
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(_lib.options(pixels=2))) == -1
    # pixels = mask lives in the workspace path's argmin march: refused (not ignored) without a workspace or without argmin
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(_lib.options(pixels=1))) == -1
    # phase (round 5: the prepass as its own enqueue): off by default, range-checked, and the two halves only exist with a workspace
    assert o.phase == 0
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(_lib.options(phase=3))) == -1
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(_lib.options(phase=1))) == -1      # no workspace in `args`
    assert L.gcfr_shadow_fwd(*args, ctypes.byref(_lib.options(phase=2))) == -1
    assert _lib.with_phase(None, 2).phase == 2 and _lib.with_phase(_lib.options(tile_w=32, pixels=1), 1).tile_w == 32
    # the copy probe validates alignment and size before launching
    assert L.gcfr_copy_probe(ctypes.c_void_p(64), ctypes.c_void_p(128), ctypes.c_size_t(24), None) == -1
    assert L.gcfr_copy_probe(ctypes.c_void_p(68), ctypes.c_void_p(128), ctypes.c_size_t(32), None) == -1


def test_product_has_no_cpu_fallback():
    import pytest
    import torch
    from geomconsistentfr_amd import RenderParams, shadow_min_distance
    from geomconsistentfr_amd._lib import GcfrError
    with pytest.raises(GcfrError):
        shadow_min_distance(torch.zeros(1, 8, 8), torch.ones(1, 8, 8), torch.ones(1, 1, 3), RenderParams())


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "geomconsistentfr_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+(oracle|c_oracle|materialised|ref_shim|postprocess_statements|normals_restatement)\b", src, re.M), f
                assert "/root/reference" not in src, f


def test_stale_library_abi_is_refused(monkeypatch):
    """A library of another ABI revision exports the same symbol names; the binding must refuse it instead of calling it
    with shifted arguments (round-2 advisor finding)."""
    import pytest
    from geomconsistentfr_amd import _lib
    L = _lib.load()
    assert L.gcfr_abi_version() == _lib.ABI_VERSION
    header = open(os.path.join(ROOT, "include", "gcfr.h")).read()
    assert int(re.search(r"#define GCFR_ABI_VERSION (\d+)", header).group(1)) == _lib.ABI_VERSION
    monkeypatch.setattr(_lib, "_LIB", None)
    monkeypatch.setattr(_lib, "ABI_VERSION", _lib.ABI_VERSION + 1)
    with pytest.raises(_lib.GcfrError, match="ABI revision"):
        _lib.load()


def test_source_hash_ignores_comments_and_sees_code(tmp_path, monkeypatch):
    """The library's content hash (stale-library check at load, `roofline.stale` in bench.py) is taken over the code, not the
    comments: editing a comment must not mark the committed profiles stale, editing a token must."""
    from geomconsistentfr_amd import build
    code = 'int a; // one\n/* two */ int b = 1; const char *s = "// kept /* kept */";\n\n   int c;   \n#define Q \'"\'\n'
    assert build._code_only(code) == 'int a;\n  int b = 1; const char *s = "// kept /* kept */";\n   int c;\n#define Q \'"\''
    assert build._code_only(code.replace("one", "another remark").replace("two", "2")) == build._code_only(code)
    assert build._code_only(code.replace("b = 1", "b = 2")) != build._code_only(code)


def test_pixels_option_reaches_gcfr_options_and_bad_values_are_refused():
    """RenderParams.pixels -> gcfr_options.pixels (host logic; no GPU needed): "mask" sets the field on a COPY of the caller's
    options and asks for the argmin plane (the option lives in the training march); anything else is refused loudly."""
    import ctypes
    import pytest
    from geomconsistentfr_amd import RenderParams, _lib
    from geomconsistentfr_amd.block import _pixels_options
    base = _lib.options(tile_w=32)
    want, opt = _pixels_options(RenderParams(pixels="mask"), False, base)
    assert want is True and opt.pixels == 1 and opt.tile_w == 32 and opt.struct_size == ctypes.sizeof(_lib.Options)
    assert base.pixels == 0                                           # the caller's struct is not modified
    want, opt = _pixels_options(RenderParams(pixels="mask"), False, None)
    assert want is True and opt.pixels == 1 and opt.ksplit == -1      # defaults + the flag
    assert _pixels_options(RenderParams(), False, base) == (False, base)
    with pytest.raises(_lib.GcfrError):
        _pixels_options(RenderParams(pixels="face"), True, None)
