"""GPU: the AUDIT build of the march -- every claim about samples it does not evaluate, checked at the moment it is made.

The bit-for-bit comparisons of this suite see a wrong claim only when it changes a minimum.  `-DGCFR_COUNTERS -DGCFR_AUDIT`
(csrc/gcfr_march.hpp) builds a march that, at every evaluation of the depth-bound test, at every early termination, for the
candidate range and wherever a lane's `any_masked` is declared irrelevant, evaluates the samples the claim speaks for plainly
(depth plane + mask: ray_sample(), the direct kernel's sample) and counts the ones that contradict it -- decisive or not.  That
is the statement the safety margins were derived for (`g > 0 => S_k >= 0.998 g^2`, T8:510-514 being a minimum over ALL samples),
and it is what tells the two mutants apart that no end-to-end test kills (profiles/r05_mutants.md: K1 = 0 and the plane term = 0
contradict hundreds of claims here -- on `pits2` and `facets` -- while changing no result).  The library is built by __graft_entry__.build() /
`tools/build_variant.sh audit -DGCFR_FAST_BUILD -DGCFR_COUNTERS -DGCFR_AUDIT` and selected with GCFR_HIP_LIB in a subprocess."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
AUDIT_LIB = os.path.join(ROOT, "geomconsistentfr_amd", "lib", "audit.so")
pytestmark = pytest.mark.gpu


def _audit(*args):
    if not os.path.exists(AUDIT_LIB):
        pytest.skip("no audit build: tools/build_variant.sh audit -DGCFR_FAST_BUILD -DGCFR_COUNTERS -DGCFR_AUDIT (or __graft_entry__.build())")
    from geomconsistentfr_amd import build as hip_build
    assert hip_build.variant_is_current(AUDIT_LIB), "lib/audit.so was built from other sources than csrc/ holds now: rebuild it (see above)"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "audit.py")] + list(args), env=dict(os.environ, GCFR_HIP_LIB=AUDIT_LIB),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-400:], r.stderr[-1500:])
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_no_claim_of_the_march_is_contradicted_on_the_soaks_random_cases():
    d = _audit("--random", "160", "--families", "none", "--seed", "5")
    assert d["violations"] == 0, d["claims_contradicted"]
    chk = d["claims_checked_lane_samples"]
    # the audit has power only if it checks: depth-bound claims in the millions, terminations and masked-sample claims too
    assert chk["depth_bound"] > 2_000_000 and chk["termination"] > 100_000 and chk["masked"] > 1_000_000, chk
    # no evaluation of the bound came near using its error budget up (1.0 = no margin left)
    assert 0.0 < d["max_share_of_Kerr_used_by_a_bound_evaluation"] < 0.5, d["max_share_of_Kerr_used_by_a_bound_evaluation"]


def test_no_claim_of_the_march_is_contradicted_on_the_directed_families():
    # (every family x 8 seeds; 60 seeds of `facets` and 24 of `pits2`: the scenes on which the audit builds of mutants 5 and 3 --
    #  the plane term and K1 of the error budget, which no end-to-end test kills -- contradict their claims, tools/mutants.py run-audit)
    d = _audit("--random", "0", "--families", "all", "--family-seeds", "8", "--more", "facets=60,pits2=24")
    assert d["violations"] == 0, (d["claims_contradicted"], {k: v for k, v in d["by_family"].items() if v["violations"]})
    assert d["claims_checked_lane_samples"]["depth_bound"] > 100_000_000, d["claims_checked_lane_samples"]
    assert d["max_share_of_Kerr_used_by_a_bound_evaluation"] < 0.5, d["by_family"]
