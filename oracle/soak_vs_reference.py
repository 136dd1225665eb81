"""TEST INFRASTRUCTURE -- authoring container only (needs /root/reference, like oracle/make_golden*.py).

Random T8 batches (B = 3, 256 x 256, 160 samples) through the IMPORTED, UNMODIFIED reference against the two oracles:
  * oracle/materialised.py   minimum_distance and argmin bit-equal to what the reference's torch.min returned (T8:514),
                             shadow weights bit-equal, shading / RGB within 1e-12 / 1e-7;
  * oracle/gcfr_oracle.c     minimum_distance within 2 f32 ulps (d = sqrt(.)/sqrt(.), T8:509: torch-CPU's vectorised sqrt is
                             not correctly rounded -- one ulp per sqrt -- the C oracle's sqrtf is), masked minima equal,
                             argmin equal wherever the distance bits agree EXCEPT ties inside that rounding (the reference's
                             own distance at the C oracle's index is within one ulp of its minimum: a handful of pixels in
                             24 M; each is re-derived from the port's (N,H,W) distances), shadow weight <= 2e-6, RGB <= 1e-6.
Input families (the regime the smooth fixtures do not reach -- round-5 verdict, missing 3):
  depth   untrained   100 x the depth head of a freshly initialised reference RelightNet (re-seeded every 10 batches)
          noise5 / noise40 / noise400   ellipsoid + Gaussian noise     uniform1500   U(-1500, 1500)     normal30  N(0, 30)
          smooth      the ellipsoid + ripple family of tests/test_oracle_vs_reference.py
  masks   random 70 % (with rectangular holes every other batch), an ellipse, all ones
  lights  random directions (z of either sign), plus -- one face per batch -- a light ON or one ulp off a boundary of
          the nine-way branch (T8:386-431; found by make_golden_rough.boundary_light), inside the image, (1,0,0), (0,0,1)

  python oracle/soak_vs_reference.py --batches 120 --out profiles/r06_oracle_vs_reference_soak.json
A 2-batch slice (--batches 2 --seed 7) runs in tests/test_oracle_vs_reference.py.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import c_oracle  # noqa: E402
import materialised as M  # noqa: E402
import ref_shim  # noqa: E402
from normals_restatement import depth_to_normals  # noqa: E402

H = W = 256
DEPTH_FAMILIES = ("untrained", "noise5", "noise40", "noise400", "uniform1500", "normal30", "smooth")


def camera(f):
    K = torch.zeros(1, 3, 3, dtype=torch.float64)
    K[:, 0, 0] = K[:, 1, 1] = f
    K[:, 2, 2] = 1.0
    K[:, 0, 2], K[:, 1, 2] = W / 2.0, H / 2.0
    return K


def snap(d):
    for _ in range(16):
        n = (np.float32(100.0) * (d / np.float32(100.0))).astype(np.float32)
        if np.array_equal(n, d):
            break
        d = n
    return d


class Inputs:
    """Seeded generator of (depth, mask, albedo, light4, tags) batches."""

    def __init__(self, seed, T8):
        self.rng = np.random.default_rng(seed)
        self.seed = seed
        self.T8 = T8
        self._net_depths = []
        self._boundary = None
        r, c = np.mgrid[0:H, 0:W]
        self.x, self.y, self.r, self.c = c - 128.0, r - 128.0, r, c

    def _ellipsoid(self):
        a, b = 80 + 20 * self.rng.random(), 95 + 25 * self.rng.random()
        return 80 * np.sqrt(np.maximum(1 - (self.x / a) ** 2 - (self.y / b) ** 2, 0)) \
            + 35 * np.exp(-(self.x ** 2 / 288 + (self.y - 12) ** 2 / 648))

    def depth(self, family):
        rng = self.rng
        if family == "untrained":
            if not self._net_depths:
                from make_golden_rough import untrained_depth
                self._net_depths = list(untrained_depth(self.T8, int(rng.integers(1, 2 ** 31))))
            return self._net_depths.pop()
        if family.startswith("noise"):
            return self._ellipsoid() + float(family[5:]) * rng.standard_normal((H, W))
        if family == "uniform1500":
            return rng.uniform(-1500, 1500, (H, W))
        if family == "normal30":
            return 30 * rng.standard_normal((H, W))
        return self._ellipsoid() + 2 * np.sin(self.c / (5.0 + 3 * rng.random())) * np.cos(self.r / 8.0)

    def mask(self, kind, holes):
        rng = self.rng
        if kind == 0:
            m = rng.random((H, W)) > 0.3
            if holes:
                for _ in range(3):
                    r0, c0 = int(rng.integers(0, 220)), int(rng.integers(0, 220))
                    m[r0:r0 + int(rng.integers(4, 60)), c0:c0 + int(rng.integers(4, 90))] = False
            return m
        if kind == 1:
            return ((self.x / (70 + 20 * rng.random())) ** 2 + (self.y / (90 + 20 * rng.random())) ** 2) < 1
        return np.ones((H, W), bool)

    def boundary_lights(self):
        if self._boundary is None:
            from make_golden_rough import boundary_light
            ulp = lambda v, s: np.nextafter(np.float32(v), np.float32(s * np.inf))
            x_lo, x_hi, y_lo, y_hi = -(W / 2.0), W - W / 2.0 - 1, 1 - H / 2.0, H / 2.0
            B = []
            for axis, val, others in [(0, x_lo, (0.55, 0.83)), (0, x_lo, (-0.2, 0.4)), (1, y_hi, (0.4, 0.9)), (1, y_hi, (-0.7, 0.3))]:
                B.append(("on", boundary_light(axis, val, others)))
                B.append(("ulp_out", boundary_light(axis, ulp(val, -1 if val < 0 else +1), others)))
                B.append(("ulp_in", boundary_light(axis, ulp(val, +1 if val < 0 else -1), others)))
            for axis, val, others in [(0, x_hi, (-0.3, 0.95)), (1, y_lo, (0.25, 0.6))]:      # 127 / -127: not reachable, nearest
                B.append(("above", boundary_light(axis, val, others, side=+1)))
                B.append(("below", boundary_light(axis, val, others, side=-1)))
            B += [("inside", np.array([0.01, -0.02, 0.9997], np.float32)), ("x_axis", np.array([1, 0, 0], np.float32)),
                  ("z_axis", np.array([0, 0, 1], np.float32)), ("neg_x_axis", np.array([-1, 0, 0], np.float32)),
                  ("y_axis_zneg", np.array([0, 1, -0.5], np.float32))]
            self._boundary = B
        return self._boundary

    def batch(self, i):
        rng = self.rng
        fams = [DEPTH_FAMILIES[(i + k * 3) % len(DEPTH_FAMILIES)] for k in range(3)]
        depth = snap(np.stack([self.depth(f) for f in fams]).astype(np.float32))
        mask = np.stack([self.mask((i + k) % 3, holes=(i % 2 == 0)) for k in range(3)]).astype(np.uint8)
        albedo = (0.15 + 0.7 * rng.random((3, 3, H, W))).astype(np.float32)
        lights = rng.standard_normal((3, 3)).astype(np.float32)
        lights[:, 2] = np.abs(lights[:, 2]) * np.where(rng.random(3) < 0.2, -1, 1)           # 20 %: z < 0 (clamped, T8:358)
        bl = self.boundary_lights()
        tag, raw = bl[i % len(bl)]
        lights[i % 3] = raw
        amb = (0.3 + 0.4 * rng.random(3)).astype(np.float32)
        return depth, mask, albedo, np.concatenate([amb[:, None], lights], 1).astype(np.float32), fams, tag


def check_batch(model, depth, mask, albedo, light4):
    """One reference forward; returns a dict of violation counts / worst errors for this batch."""
    logits = np.log(albedo.astype(np.float64) / (1 - albedo)).astype(np.float32)
    ref_shim.inject(model, torch.from_numpy(depth / np.float32(100.0))[:, None], torch.from_numpy(logits),
                    torch.from_numpy(light4).view(3, 1, 1, 4))
    with torch.no_grad(), ref_shim.capture_min() as cap:
        out = model(torch.zeros(3, H, W, 3), 200, camera(1570.0), torch.from_numpy(mask.astype(np.float64))[..., None])
    assert np.array_equal(out[1].numpy()[:, 0], depth)
    md_ref, am_ref = np.stack(cap.values), np.stack(cap.indices)
    albedo_used = out[0].numpy()
    res = {}
    # ---- materialised port ----
    p = M.BlockParams()
    n = depth_to_normals(torch.from_numpy(depth)[:, None] + 1610.0, camera(1570.0))
    n = torch.cat([n[:, 0:1], -n[:, 1:2], n[:, 2:3]], 1)
    with torch.no_grad():
        o = M.render_block(torch.from_numpy(depth)[:, None], torch.from_numpy(albedo_used), torch.from_numpy(light4[:, 1:4]),
                           torch.from_numpy(light4[:, 0]), n, torch.from_numpy(mask), p)
        _, pt = M.light_points(torch.from_numpy(light4[:, 1:4]), p)
        bad_v = bad_i = 0
        for b in range(3):
            v, idx = M.min_distance_one(torch.from_numpy(depth[b]), torch.from_numpy(mask[b]), pt[b], p)
            bad_v += int((v.numpy() != md_ref[b]).sum())
            bad_i += int((idx.numpy() != am_ref[b]).sum())
    res["mat_value_mismatch"] = bad_v
    res["mat_index_mismatch"] = bad_i
    res["mat_w_mismatch"] = int((o["shadow_mask_weights"].numpy() != out[2].numpy()).sum())
    res["mat_full_err"] = float(np.abs(o["full_shading"].numpy() - out[4].numpy()).max())
    res["mat_rgb_err"] = float(np.abs(o["rendered_images"].numpy() - out[5].numpy()).max())
    # ---- C oracle ----
    unit, ptc = c_oracle.light_prep(light4[:, 1:4], clamp_z_min=0.0)
    res["c_unit_mismatch"] = int((unit != out[6].numpy().reshape(3, 3)).sum())
    md, am = c_oracle.shadow_min_distance(depth, mask, ptc[:, None, :], c_oracle.sample_table())
    md, am = md[:, 0], am[:, 0]
    lit = md_ref < 1e5
    res["c_lit_mismatch"] = int((lit != (md < 1e5)).sum())
    both = lit & (md < 1e5)
    res["c_masked_value_mismatch"] = int((md[~lit] != md_ref[~lit]).sum())
    rel = np.abs(md[both].astype(np.float64) - md_ref[both]) / np.maximum(np.abs(md_ref[both]), 1e-30)
    res["c_rel_err"] = float(rel.max()) if rel.size else 0.0
    ulps = np.abs(md[both].view(np.int32).astype(np.int64) - md_ref[both].view(np.int32).astype(np.int64))
    res["c_max_ulps"] = int(ulps.max()) if ulps.size else 0
    res["c_value_biteq_frac"] = float((md[both] == md_ref[both]).mean()) if both.any() else 1.0
    same = both & (md == md_ref)
    differ = same & (am != am_ref)
    res["c_index_mismatch_where_bits_agree"] = int(differ.sum())
    # ... each of them must be a TIE inside the reference's own rounding: the reference's distance at the C oracle's index is
    # within one ulp of the reference's minimum (its vectorised sqrt put two samples the C oracle separates on the same value, or
    # the other way round; first index wins, T8:514).  Anything else is a violation.
    unexplained = 0
    for b in np.unique(np.nonzero(differ)[0]):
        with torch.no_grad():
            _, _, d_all = M.min_distance_one(torch.from_numpy(depth[b]), torch.from_numpy(mask[b]), pt[b], p, return_all=True)
        d_all = d_all.numpy()
        for r_, c_ in zip(*np.nonzero(differ[b])):
            at_c = np.float32(d_all[am[b, r_, c_], r_, c_])
            ulps = abs(int(at_c.view(np.int32)) - int(md_ref[b, r_, c_].view(np.int32)))
            unexplained += int(ulps > 1)
    res["c_index_mismatch_unexplained"] = unexplained
    res["c_index_eq_frac"] = float((am[both] == am_ref[both]).mean()) if both.any() else 1.0
    sh = c_oracle.shade(n.numpy(), depth, albedo_used, ptc[:, None, :], light4[:, :1], md[:, None])
    res["c_w_err"] = float(np.abs(sh["shadow_w"][:, 0] - out[2].numpy()).max())
    res["c_full_err"] = float(np.abs(sh["full_shading"][:, 0] - out[4].numpy()).max())
    res["c_rgb_err"] = float(np.abs(sh["rendered"][:, 0] - out[5].numpy()).max())
    res["lit_frac"] = float(lit.mean())
    return res


def violations(r):
    v = []
    for k in ("mat_value_mismatch", "mat_index_mismatch", "mat_w_mismatch", "c_unit_mismatch", "c_lit_mismatch",
              "c_masked_value_mismatch", "c_index_mismatch_unexplained"):
        if r[k]:
            v.append(k)
    if r["mat_full_err"] > 1e-12 or r["mat_rgb_err"] > 1e-7:
        v.append("mat_shading")
    if r["c_max_ulps"] > 2:
        v.append("c_max_ulps")
    if r["c_w_err"] > 2e-6 or r["c_full_err"] > 1e-6 or r["c_rgb_err"] > 1e-6:
        v.append("c_shading")
    return v


def run(batches, seed, verbose=True):
    T8 = ref_shim.load("T8")
    model = T8.RelightNet()
    gen = Inputs(seed, T8)
    t0 = time.time()
    rows, n_viol = [], 0
    for i in range(batches):
        depth, mask, albedo, light4, fams, tag = gen.batch(i)
        r = check_batch(model, depth, mask, albedo, light4)
        r.update(batch=i, depth_families=fams, boundary_light=tag, violations=violations(r))
        n_viol += len(r["violations"])
        rows.append(r)
        if verbose:
            print("batch %3d %-34s %-8s lit %.2f  C: ulps %d rel %.2e biteq %.4f idx %.5f w %.1e  viol %s  (%.0fs)" % (
                i, "/".join(fams), tag, r["lit_frac"], r["c_max_ulps"], r["c_rel_err"], r["c_value_biteq_frac"], r["c_index_eq_frac"],
                r["c_w_err"], r["violations"], time.time() - t0), flush=True)
    summary = dict(
        what="imported unmodified reference (T8 forward, B=3, 256x256x160) vs oracle/materialised.py and oracle/gcfr_oracle.c",
        batches=batches, faces=3 * batches, seed=seed, pixels=3 * batches * H * W, violations=n_viol,
        materialised=dict(value_mismatches=sum(r["mat_value_mismatch"] for r in rows),
                          index_mismatches=sum(r["mat_index_mismatch"] for r in rows),
                          w_mismatches=sum(r["mat_w_mismatch"] for r in rows),
                          worst_full_shading_err=max(r["mat_full_err"] for r in rows),
                          worst_rgb_err=max(r["mat_rgb_err"] for r in rows)),
        c_oracle=dict(worst_ulps=max(r["c_max_ulps"] for r in rows), worst_relative_err=max(r["c_rel_err"] for r in rows),
                      min_biteq_frac=min(r["c_value_biteq_frac"] for r in rows),
                      min_index_eq_frac=min(r["c_index_eq_frac"] for r in rows),
                      index_mismatches_where_bits_agree=sum(r["c_index_mismatch_where_bits_agree"] for r in rows),
                      index_mismatches_not_a_one_ulp_tie=sum(r["c_index_mismatch_unexplained"] for r in rows),
                      lit_mismatches=sum(r["c_lit_mismatch"] for r in rows),
                      worst_w_err=max(r["c_w_err"] for r in rows), worst_full_shading_err=max(r["c_full_err"] for r in rows),
                      worst_rgb_err=max(r["c_rgb_err"] for r in rows)),
        depth_families=sorted({f for r in rows for f in r["depth_families"]}),
        boundary_lights=sorted({r["boundary_light"] for r in rows}),
        seconds=round(time.time() - t0, 1))
    return summary, rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", type=int, default=120)
    ap.add_argument("--seed", type=int, default=606)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    summary, rows = run(a.batches, a.seed)
    print(json.dumps(summary))
    if a.out:
        with open(a.out, "w") as f:
            json.dump(dict(summary=summary, batches=rows), f, indent=1)
    sys.exit(1 if summary["violations"] else 0)


if __name__ == "__main__":
    main()
