"""CPU: bench.py's host-side rules -- how many fenced regions a short run gets, the FFHQ fixture workload, the spec-rate
pricing of the committed instruction mix."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_short_regions_are_repeated_and_long_ones_are_not():
    assert bench.regions_needed(3000, 0.07) == 1                   # 210 ms: one region is a sample
    assert bench.regions_needed(20, 0.05) == 200                   # the driver's --steps 20: 1 ms regions, capped at 200
    assert bench.regions_needed(1000, 0.05) == 40                  # 50 ms regions: ~2 s in total
    assert bench.regions_needed(20, 9.0) == 25                     # 180 ms regions: the floor of 25
    bench.FORCED_REGIONS = 1
    try:
        assert bench.regions_needed(20, 0.05) == 1                 # --regions 1 (profiling scripts)
    finally:
        bench.FORCED_REGIONS = 0


def test_ffhq_fixture_workload_is_the_three_faces_tiled_and_mirrored():
    depth, mask, albedo, normals, light, amb = bench.ffhq_faces(8, first=0)
    inp = np.load(os.path.join(ROOT, "tests", "golden", "inputs.npz"))
    assert depth.shape == (8, 256, 256) and mask.shape == (8, 256, 256) and mask.dtype == np.uint8
    np.testing.assert_array_equal(depth[0], inp["depths"][1])
    np.testing.assert_array_equal(mask[2], inp["masks"][4])
    np.testing.assert_array_equal(depth[3], inp["depths"][1][:, ::-1])          # second pass through the three: mirrored
    np.testing.assert_array_equal(depth[6], inp["depths"][1])
    np.testing.assert_allclose(np.linalg.norm(normals, axis=1), 1.0, atol=1e-5)
    assert 0.25 < mask.mean() < 0.45                                            # skin masks cover about a third of the frame
    np.testing.assert_array_equal(light[:3], bench.LIGHTS18[:3])


def test_spec_rate_pricing_of_the_committed_mix():
    """frac_spec: the committed SQ_INSTS_VALU_* mix at the guide's issue rates (2 cycles per wave64 f32 / int op, 4 per f64 /
    cvt, quarter-rate transcendentals) -- every class of the summary is priced, and the total sits below the measured-cost total."""
    pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_summary.json")))
    fwd = pm["kernels"]["fwd"]["valu"]
    assert set(fwd["by_class"]) == set(bench.SPEC_CYCLES)
    spec = sum(fwd["by_class"][k] * c for k, c in bench.SPEC_CYCLES.items())
    assert 0.6 * fwd["issue_cycles_per_launch"] < spec < fwd["issue_cycles_per_launch"]
    assert abs(sum(fwd["by_class"].values()) - fwd["insts_per_launch"]) < 1.0
    assert pm.get("library_srchash")                                            # what bench.py's `roofline.stale` compares


def test_compact_line_keeps_the_contract_keys_and_drops_the_prose():
    """bench.compact(): long `note`s and per-run lists go, every number and the contract's strings (`sample`, `workload`) stay."""
    line = {"metric": "ray_steps_per_sec", "value": 1.0, "config": {"workload": "w" * 300, "faces_per_gpu": 8},
            "roofline": {"frac": 0.3, "note": "n" * 500, "hbm": {"frac": 2.7, "note": "short"}},
            "cpu_baseline": {"value": 8e6, "unit": "ray-steps/s", "cores": 8, "kind": "port", "sample": "s" * 100,
                             "forward_by_threads": {"8": {"runs_s": [1.0, 2.0]}}, "forward_backward": {"value": 4e6, "runs_s": [3.0]}}}
    c = bench.compact(line)
    assert c["roofline"] == {"frac": 0.3, "hbm": {"frac": 2.7, "note": "short"}}
    assert c["cpu_baseline"]["sample"] == "s" * 100 and "forward_by_threads" not in c["cpu_baseline"]
    assert c["cpu_baseline"]["forward_backward"] == {"value": 4e6}
    assert c["config"]["workload"].endswith("...") and len(c["config"]["workload"]) == 240 and c["config"]["faces_per_gpu"] == 8
    assert json.loads(json.dumps(c)) == c
