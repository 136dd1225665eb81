"""GPU: directed tests of the march's safety margins (csrc/gcfr_mutants.hpp).

The march skips samples on the strength of hand-derived margins; the reference (T8:510-514) takes a minimum over ALL samples,
so every skipped sample is a claim.  Round 5 built one library per margin with that margin removed or inverted
(`-DGCFR_MUT=<n>`) and ran the `-m gpu` suite against each: 17 of 28 mutants SURVIVED the suite of round 4.  The scenes below
are the answer: each family (tests/margin_scenes.py) is built from the geometry of one mechanism -- rays parallel to a plane a
hair above it, a light below its pixels, a wall one cell behind the ray, distances around the masked value 1e6, a sample at an
integral coordinate (found by search), a sample table at the edge of what the prepass accepts ... -- and the seeds listed here
are the ones on which the mutant named beside them gives other bits than the unmutated build (tools/mutant_hunt.py).  The
test itself knows nothing of mutants: it compares the product, bit for bit, with the C oracle.  profiles/r05_mutants.md has
the table mutant -> first failing test."""
import ctypes
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")

# (family, seed, the mutants of csrc/gcfr_mutants.hpp this scene is known to kill)
KILLERS = [
    ("diamond", 2, [2, 15, 17, 24]), ("diamond", 7, [2]),                 # octagon inflation; pixels = mask
    ("integral", 0, [10]), ("integral", 1, [10, 25]), ("integral", 3, [25]),   # the sampled zero: gap0, the cap's 0
    ("descending", 0, [11, 15, 26, 27]), ("descending", 1, [11]), ("descending", 5, [6]),   # "the ray is still rising"
    ("million", 1, [23, 24]), ("negative", 5, [23]), ("parallel", 36, [23]), ("parallel", 38, [23]),   # Kerr
    ("table", 1, [1]), ("table", 29, [1]),                                 # box inflation under an uneven table
    ("cliffs", 0, [29]), ("cliffs", 8, [16]),
    ("level2", 0, [7, 8]), ("level2", 1, [7, 8]),                          # the termination tests' 0.2 % (main loop, trailing loop)
    ("million3", 0, [9, 18]), ("million3", 10, [9]), ("million3", 4, [18]),   # safeS: "certainly below the masked 1e6"
    ("cliffs2", 18, [22]), ("cliffs2", 63, [22]),                          # the bounds grid's stride: a group's reach + 3 cells
    ("parallel3", 0, [4, 23]), ("parallel3", 1, [4]),                      # K2 r: the f32 noise of the distance far from zero
    ("pits2", 0, [23]), ("level", 12, [23]),
    ("wrap_column", 0, [14]), ("wrap_column", 1, []),                      # the horizon tables' wrap partners: a ramped wall in the
    ("wrap_last_sample", 0, [13]), ("wrap_last_sample", 1, [13]),          # wrap-partner column / row decides late samples (round 6);
]                                                                          # a table reaching t = 1: the PREFIX tables' wrap partner


def _march(sc, want_argmin, pixels=0):
    from geomconsistentfr_amd import _lib
    L_ = _lib.load()
    depth, mask, pt, tt = [torch.from_numpy(np.ascontiguousarray(a)).to(DEV) for a in (sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"])]
    B, H, W = depth.shape
    L = pt.shape[1]
    md = torch.empty((B, L, H, W), dtype=torch.float32, device=DEV)
    am = torch.empty((B, L, H, W), dtype=torch.int32, device=DEV) if want_argmin else None
    nb = int(L_.gcfr_shadow_workspace_bytes(B, H, W))
    ws = torch.empty(nb, dtype=torch.uint8, device=DEV)
    opt = _lib.options(ksplit=0, pixels=pixels)           # the grid schedule: bounds skip, trailing loop, horizon tables
    _lib.check(L_.gcfr_shadow_fwd(depth.data_ptr(), mask.data_ptr(), mask.shape[0], pt.data_ptr(), B, L, H, W, tt.numel(), tt.data_ptr(), 0.0,
                                  None, md.data_ptr(), am.data_ptr() if am is not None else None, ws.data_ptr(), nb, None,
                                  ctypes.byref(opt)), "gcfr_shadow_fwd")
    torch.cuda.synchronize()
    return md.cpu().numpy(), (am.cpu().numpy() if am is not None else None)


@pytest.mark.parametrize("family,seed,kills", KILLERS, ids=["%s-%d" % (f, s) for f, s, _ in KILLERS])
def test_margin_scene_matches_the_oracle_bit_for_bit(family, seed, kills):
    import c_oracle
    import margin_scenes as MS
    sc = MS.FAMILIES[family](seed)
    md_o, am_o = c_oracle.shadow_min_distance(sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"])
    lit = md_o < 1e5
    for want in (True, False):
        md, am = _march(sc, want)
        bad = np.argwhere(md.view(np.int32) != md_o.view(np.int32))
        assert bad.size == 0, (family, seed, want, len(bad), bad[:3].tolist(), "mutants this scene kills: %s" % (kills,))
        if want:
            assert np.array_equal(am[lit], am_o[lit]), (family, seed)
    if sc.get("pixels_mask"):                            # pixels = mask: the pixels inside the mask keep their bits, the others carry 1e6
        md, am = _march(sc, True, pixels=1)
        own = np.broadcast_to(sc["mask"][:, None] != 0, md.shape)
        assert np.array_equal(md[own].view(np.int32), md_o[own].view(np.int32)) and np.all(md[~own] == 1e6)


def test_integral_coordinate_scenes_really_sample_zero():
    """The search of family `integral` is the point of it: at the critical pixel the reference's bilinear weights are both zero at one
    sample, the sampled depth is 0 whatever the (negative) depth map holds, and that sample is the minimum -- in the ORACLE, i.e. in
    the reference's arithmetic; the march must find it although no depth-bounds tile and no depth maximum knows of a 0."""
    import c_oracle
    import margin_scenes as MS
    for seed in (0, 1):
        sc = MS.FAMILIES["integral"](seed)
        assert sc["found"] >= 1
        md_o, am_o = c_oracle.shadow_min_distance(sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"])
        for want in (True, False):
            md, am = _march(sc, want)
            assert np.array_equal(md.view(np.int32), md_o.view(np.int32)), seed
        for b, r, c in sc["critical"]:
            assert md_o[b, 0, r, c] < 1e-3 < md_o[b, 0, r, c + 1]        # the neighbour's ray never sees the zero
            assert 8 <= am_o[b, 0, r, c] < 150
