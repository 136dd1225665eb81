#!/usr/bin/env python3
"""The AUDIT build of the march (GPU): every claim about samples it does not evaluate, checked against those samples.

    tools/build_variant.sh audit -DGCFR_FAST_BUILD -DGCFR_COUNTERS -DGCFR_AUDIT           (CPU)
    GCFR_HIP_LIB=geomconsistentfr_amd/lib/audit.so tools/audit.py [--random 400] [--families all] [--family-seeds 6] [--seed 0]

csrc/gcfr_march.hpp (-DGCFR_AUDIT): at every evaluation of the depth-bound test, at every early termination, for the candidate range
and wherever a lane's `any_masked` is declared irrelevant, the samples the claim speaks for are evaluated plainly (ray_sample(): depth
plane + mask, no workspace, no bounds) and compared with the claim -- whether or not the claim decided anything.  The scenes are the
soak's random cases (tools/soak_parity.py: sizes, depth scales, light distances) and the directed families of tests/margin_scenes.py;
both march kernels (with / without argmin) run each.  Prints one JSON object: claims checked, contradicted (must be 0), and the largest
share of the error budget Kerr a depth-bound evaluation used up (1.0 = a bound with no margin left).

With a mutant's audit build (tools/mutants.py build-audit) the same scenes show whether the margin the mutant removes is needed for
the claims to HOLD, not only whether its removal changes a result (profiles/r05_mutants.md, column "audit")."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
AUDIT_KEYS = ("audit_bound_checks", "audit_bound_violations", "audit_term_checks", "audit_term_violations", "audit_masked_checks",
              "audit_masked_violations", "audit_safe_violations", "audit_max_use")


class Auditor:
    def __init__(self, knobs=None):
        from geomconsistentfr_amd import _lib
        self.knobs = dict(knobs or {})
        self._lib, self.L_ = _lib, _lib.load()
        self.version = self.L_.gcfr_version().decode()
        if "+audit" not in self.version:
            raise SystemExit("this is not an audit build (%s): GCFR_HIP_LIB=geomconsistentfr_amd/lib/audit.so, see the docstring" % self.version)
        self.dev = torch.device("cuda:0")
        self.tot = dict.fromkeys(AUDIT_KEYS, 0)
        self.tot.update(bound_tests=0, groups_visited=0, tiles=0, launches=0)
        self.by_family = {}

    def march(self, depth, mask, light_pt, t_table, want_argmin, pixels=0, tag="random"):
        """one launch of gcfr_shadow_fwd (grid schedule) with a fresh counter array; its audit tallies are added to the totals"""
        _lib, L_, dev = self._lib, self.L_, self.dev
        t = lambda a_: torch.from_numpy(np.ascontiguousarray(a_)).to(dev)
        d_depth, d_mask, d_pt, d_tt = t(depth), t(mask), t(light_pt), t(t_table)
        B, H, W = depth.shape
        L = light_pt.shape[1]
        md = torch.empty((B, L, H, W), dtype=torch.float32, device=dev)
        am = torch.empty((B, L, H, W), dtype=torch.int32, device=dev) if want_argmin else None
        nb = int(L_.gcfr_shadow_workspace_bytes(B, H, W))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        tw = self.knobs.get("tile_w", 0) or 16
        n_tiles = B * L * ((W + tw - 1) // tw) * ((H + 64 // tw - 1) // (64 // tw))
        counters = torch.zeros(_lib.N_COUNTERS + 4 * n_tiles, dtype=torch.int64, device=dev)
        opt = _lib.options(**dict(dict(ksplit=0), **self.knobs), pixels=pixels, counters=counters.data_ptr())
        _lib.check(L_.gcfr_shadow_fwd(d_depth.data_ptr(), d_mask.data_ptr(), d_mask.shape[0], d_pt.data_ptr(), B, L, H, W, d_tt.numel(),
                                      d_tt.data_ptr(), 0.0, None, md.data_ptr(), am.data_ptr() if am is not None else None, ws.data_ptr(), nb,
                                      None, ctypes.byref(opt)), "gcfr_shadow_fwd")
        torch.cuda.synchronize()
        c = dict(zip(_lib.COUNTER_NAMES, counters[:_lib.N_COUNTERS].cpu().tolist()))
        fam = self.by_family.setdefault(tag, dict.fromkeys(AUDIT_KEYS, 0))
        for acc in (self.tot, fam):
            for k in AUDIT_KEYS:
                acc[k] = max(acc[k], c[k]) if k == "audit_max_use" else acc[k] + c[k]
        for k in ("bound_tests", "groups_visited", "tiles"):
            self.tot[k] += c[k]
        self.tot["launches"] += 1
        return md, am

    def random_cases(self, n_cases, seed):
        """the soak's cases (tools/soak_parity.py run_soak: same sizes, depth scales and light distances), batches of 8"""
        import soak_parity as SP
        import c_oracle
        from geomconsistentfr_amd import RenderParams
        rng = np.random.default_rng(seed)
        sizes = [(64, 64, 48), (96, 128, 80), (130, 70, 37), (128, 128, 160), (256, 256, 160)]
        B = 8
        for it in range(n_cases // B):
            H, W, N = sizes[it % len(sizes)]
            cases = [SP.random_case(rng, H, W) for _ in range(B)]
            depth = np.stack([c[0] for c in cases])
            mask = np.stack([c[1] for c in cases])
            lights = np.stack([c[2] for c in cases])
            scale = [1.0, 1.0, 0.01, 12.0, 300.0][it % 5] if it % 3 == 0 else 1.0
            depth = (depth * np.float32(scale)).astype(np.float32)
            ld = [4013.0, 4013.0, 60.0, 500.0, 1.0e5, 30.0][(it // 5) % 6]
            _, pt = c_oracle.light_prep(lights, clamp_z_min=0.0, light_distance=ld)
            tt = c_oracle.sample_table(0.025, 0.8 / N, N)
            for want in (True, False):
                self.march(depth, mask, pt[:, None, :].astype(np.float32), tt, want, tag="random")

    def config5(self, n_faces, seed):
        """random faces at BASELINE configs[4]'s shape: 512 x 512, 320 samples, 18 lights per face in one launch"""
        import soak_parity as SP
        import c_oracle
        rng = np.random.default_rng(seed)
        tt = c_oracle.sample_table(0.025, 0.8 / 320, 320)
        for _ in range(n_faces):
            depth, mask, _l = SP.random_case(rng, 512, 512)
            lights = rng.standard_normal((18, 3)).astype(np.float32)
            lights[:, 2] = np.abs(lights[:, 2]) * rng.choice([1.0, 0.1])
            _, pt = c_oracle.light_prep(lights, clamp_z_min=0.0)
            for want in (True, False):
                self.march(depth[None], mask[None], pt[None].astype(np.float32), tt, want, tag="config5")

    def families(self, names, n_seeds, more=None):
        import margin_scenes as MS
        for name in names:
            for seed in range(max(n_seeds, (more or {}).get(name, 0))):
                sc = MS.FAMILIES[name](seed)
                for want in (True, False):
                    self.march(sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"], want, tag=name)
                if sc.get("pixels_mask"):
                    self.march(sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"], True, pixels=1, tag=name)

    def report(self):
        t = self.tot
        viol = t["audit_bound_violations"] + t["audit_term_violations"] + t["audit_masked_violations"] + t["audit_safe_violations"]
        return {"library": self.version, "launches": t["launches"], "tiles": t["tiles"], "bound_tests_wave_level": t["bound_tests"],
                "claims_checked_lane_samples": {"depth_bound": t["audit_bound_checks"], "termination": t["audit_term_checks"],
                                                "masked": t["audit_masked_checks"]},
                "claims_contradicted": {"depth_bound": t["audit_bound_violations"], "termination": t["audit_term_violations"],
                                        "masked": t["audit_masked_violations"], "distance_below_masked_value": t["audit_safe_violations"]},
                "violations": viol, "max_share_of_Kerr_used_by_a_bound_evaluation": t["audit_max_use"] / 1000.0,
                "by_family": {k: {"violations": v["audit_bound_violations"] + v["audit_term_violations"] + v["audit_masked_violations"] + v["audit_safe_violations"],
                                  "bound_checks": v["audit_bound_checks"], "max_share_of_Kerr": v["audit_max_use"] / 1000.0} for k, v in self.by_family.items()}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--random", type=int, default=400, help="random soak cases (multiples of 8)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--families", type=str, default="all", help="'all', 'none' or a comma list of tests/margin_scenes.py families")
    ap.add_argument("--family-seeds", type=int, default=6)
    ap.add_argument("--more", type=str, default="facets=60,pits2=24",
                    help="families that get more seeds: the ones on which the single terms of Kerr are needed for a claim to hold "
                         "(mutant 5, the plane term: steep planar facets under level light; mutant 3, K1: plateaus with pits under an overhead light)")
    ap.add_argument("--config5", type=int, default=0, help="N random faces at configs[4]'s shape (512 x 512 x 320, 18 lights)")
    ap.add_argument("--tune", type=str, default="", help="gcfr_options knobs, e.g. tile_w=8,group=2 or lds_stage=1 (needs an audit build "
                                                          "of every march unit: without -DGCFR_FAST_BUILD)")
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    import margin_scenes as MS
    knobs = {k: int(v) for k, v in (kv.split("=") for kv in a.tune.split(",") if kv)}
    au = Auditor(knobs)
    t0 = time.time()
    if a.random:
        au.random_cases(a.random, a.seed)
    if a.config5:
        au.config5(a.config5, a.seed)
    names = [] if a.families == "none" else (sorted(MS.FAMILIES) if a.families == "all" else a.families.split(","))
    more = {k: int(v) for k, v in (kv.split("=") for kv in a.more.split(",") if kv)}
    au.families(names, a.family_seeds, {k: v for k, v in more.items() if k in names})
    r = au.report()
    r.update(knobs=knobs, random_cases=(a.random // 8) * 8, config5_faces=a.config5, seed=a.seed, families=names, family_seeds=a.family_seeds, more_seeds=more,
             seconds=round(time.time() - t0, 1))
    print(json.dumps(r))
    if a.out:
        with open(os.path.join(ROOT, a.out), "w") as f:
            json.dump(r, f, indent=1)


if __name__ == "__main__":
    main()
