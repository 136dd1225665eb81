"""CPU: the mutation-testing machinery of the march (csrc/gcfr_mutants.hpp) is consistent with itself and invisible in the product.

* every GCFR_M(n, ...) site in the sources has a row in the header's table and its GCFR_MUT_PAIR_n_n definition, and vice versa;
* with GCFR_MUT undefined, GCFR_M(n, mutant, product) is the product's tokens (preprocessed with the host compiler: plain token
  selection, no code generation involved), and with -DGCFR_MUT=n exactly site n flips;
* the scenes tests/test_gpu_margins.py pins exist and are deterministic (numpy only);
* tools/census.py's stage markers cover the march's tile function in order."""
import os
import re
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "geomconsistentfr_amd", "csrc")
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _header():
    return open(os.path.join(CSRC, "gcfr_mutants.hpp")).read()


def test_every_mutant_site_has_a_table_row_and_a_pair_definition():
    h = _header()
    rows = {int(m.group(1)) for m in re.finditer(r"^// +(\d+)  \S", h, re.M)}
    pairs = {int(m.group(1)) for m in re.finditer(r"#define GCFR_MUT_PAIR_(\d+)_\1 ~, 1", h)}
    sites = set()
    for f in os.listdir(CSRC):
        if f == "gcfr_mutants.hpp":
            continue
        sites |= {int(m.group(1)) for m in re.finditer(r"GCFR_M\((\d+),", open(os.path.join(CSRC, f)).read())}
    assert rows == pairs == sites, (sorted(rows ^ pairs), sorted(rows ^ sites))
    assert len(rows) >= 14


def test_token_selection_is_the_products_text_unless_the_mutant_is_named(tmp_path):
    src = tmp_path / "probe.cpp"
    src.write_text('#include "gcfr_mutants.hpp"\n'
                   "A GCFR_M(6, 1.002f, 0.998f) B GCFR_M(26, + 1, ) C GCFR_M(16, false &&, ) D GCFR_M(13, 0, s_col[W - 1]) E\n")

    def pre(*defs):
        r = subprocess.run(["g++", "-E", "-P", "-I", CSRC] + list(defs) + [str(src)], capture_output=True, text=True, check=True)
        return " ".join(r.stdout.split())
    assert pre() == "A 0.998f B C D s_col[W - 1] E"
    assert pre("-DGCFR_MUT=0") == pre()
    assert pre("-DGCFR_MUT=6") == "A 1.002f B C D s_col[W - 1] E"
    assert pre("-DGCFR_MUT=26") == "A 0.998f B + 1 C D s_col[W - 1] E"
    assert pre("-DGCFR_MUT=16") == "A 0.998f B C false && D s_col[W - 1] E"
    assert pre("-DGCFR_MUT=13") == "A 0.998f B C D 0 E"
    assert pre("-DGCFR_MUT=99") == pre()        # an unknown number selects nothing


def test_the_pinned_scenes_exist_and_are_deterministic():
    import margin_scenes as MS
    text = open(os.path.join(ROOT, "tests", "test_gpu_margins.py")).read()
    killers = re.findall(r'\("(\w+)", (\d+), \[([\d, ]+)\]\)', text)
    assert len(killers) >= 25
    killed = set()
    for fam, seed, kills in killers:
        assert fam in MS.FAMILIES, fam
        killed |= {int(k) for k in kills.split(",")}
    for fam in sorted({k[0] for k in killers})[:6]:                 # (generation is cheap; a few families twice)
        a, b = MS.FAMILIES[fam](3), MS.FAMILIES[fam](3)
        for key in ("depth", "mask", "light_pt", "t_table"):
            assert np.array_equal(a[key], b[key]), (fam, key)
        assert a["depth"].dtype == np.float32 and a["mask"].dtype == np.uint8 and a["t_table"].dtype == np.float64
    # every mutant the GPU scenes are responsible for is named by at least one of them
    assert {2, 4, 7, 8, 9, 10, 11, 17, 18, 22, 23, 25} <= killed


def test_integral_coordinate_search_finds_exact_hits():
    """family `integral`: the f64 pipeline ((0 + t_k dx) + W/2) - 0.0001 really lands on an integer for the (k, dx) it returns"""
    import margin_scenes as MS
    for seed in (0, 1, 2):
        sc = MS.FAMILIES["integral"](seed)
        assert sc["found"] >= 1
        W = sc["depth"].shape[2]
        dx = np.float64(sc["light_pt"][0, 0, 0])
        hit = [k for k, t in enumerate(sc["t_table"]) if ((0.0 + t * dx) + W / 2.0) - 0.0001 == np.floor(((0.0 + t * dx) + W / 2.0) - 0.0001)]
        assert hit, seed


def test_census_markers_cover_the_tile_function_in_order():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import census
    maps = census.stage_maps()
    assert set(maps) >= {"gcfr_march.hpp", "gcfr_device.hpp"}
    names = [n for _, n in maps["gcfr_march.hpp"]]
    order = ["tile set-up", "end point", "give-up test", "candidate range", "bounds set-up", "loop: bounds record fetch", "loop: trailing loop",
             "epilogue: distance finish", "epilogue: pixel re-derivation", "epilogue: shading call"]
    pos = [next(i for i, n in enumerate(names) if n.startswith(o)) for o in order]
    assert pos == sorted(pos), list(zip(order, pos))
    assert census.classify("v_fma_f64")[1:] == ("f64", 4) and census.classify("v_rcp_f32_e32")[1:] == ("trans_f32", 8)
    assert census.classify("buffer_load_dwordx4")[0] == "vmem" and census.classify("s_cbranch_scc1")[0] == "branch"


def test_every_scene_family_is_well_formed_and_deterministic():
    """what tools/audit.py and tools/mutant_hunt.py iterate over: depth (B,H,W) f32, mask (B,H,W) u8, light points (B,L,3) f32, an
    increasing f64 sample table -- and the same arrays for the same seed"""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import margin_scenes as MS
    assert {"facets", "pits2", "wrap_edge", "integral", "million3", "level2"} <= set(MS.FAMILIES)
    for name, fam in sorted(MS.FAMILIES.items()):
        a, b = fam(1), fam(1)
        B, H, W = a["depth"].shape
        assert a["depth"].dtype == np.float32 and a["mask"].dtype == np.uint8 and a["mask"].shape == (B, H, W), name
        assert a["light_pt"].dtype == np.float32 and a["light_pt"].shape[0] == B and a["light_pt"].shape[2] == 3, name
        tt = a["t_table"]
        assert tt.dtype == np.float64 and tt.ndim == 1 and np.all(np.diff(tt) > 0), name
        for k in ("depth", "mask", "light_pt", "t_table"):
            assert np.array_equal(a[k], b[k]), (name, k)


def test_a_variant_library_is_current_only_with_the_hash_of_todays_sources(tmp_path):
    """geomconsistentfr_amd/build.py: every compile_and_link() leaves <name>.srchash beside the library; tests that run a variant
    (tests/test_gpu_audit.py) refuse one built from other sources"""
    from geomconsistentfr_amd import build as hip_build
    lib = tmp_path / "variant.so"
    lib.write_bytes(b"")
    assert not hip_build.variant_is_current(str(lib))                     # no hash beside it
    (tmp_path / "variant.srchash").write_text("0" * 64 + "\n" + "0" * 64 + "\n-DX\n")
    assert not hip_build.variant_is_current(str(lib))                     # another source
    (tmp_path / "variant.srchash").write_text(hip_build.source_hash() + "\n" + hip_build.full_bytes_hash() + "\n-DX\n")
    assert hip_build.variant_is_current(str(lib))
