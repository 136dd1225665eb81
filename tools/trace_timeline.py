#!/usr/bin/env python3
"""Wave timeline of one march launch (GPU box, counting build): where does the time of a single launch go?

Every tile's wave records entry / exit on the constant 100 MHz clock (s_memrealtime), its shader-cycle count, the
SIMD / CU / XCC it ran on and how many groups / bodies it executed.  This script launches the bench workload once per
requested schedule and prints: kernel span, tile-duration quantiles, how many waves are in flight over time, the busy
time of the busiest and the median SIMD, and the time at which 50 / 90 / 99 % of the tiles are done -- the numbers
that tell a tail problem (few long waves at the end) from an imbalance problem (some SIMDs overloaded throughout)
from a rate problem (every wave slow).

usage: GCFR_HIP_LIB=geomconsistentfr_amd/lib/counters.so python tools/trace_timeline.py [--faces 8]
       [--tune "tile_w=16" --tune "tile_w=8,group=2" ...] [--out gpurun_out/x.npz]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from geomconsistentfr_amd import RenderParams, _lib  # noqa: E402
from geomconsistentfr_amd import block as R  # noqa: E402


def analyse(rec, label):
    t0, t1, cyc, meta = rec[:, 0], rec[:, 1], rec[:, 2], rec[:, 3]
    ok = t1 > 0
    t0, t1, cyc, meta = t0[ok], t1[ok], cyc[ok], meta[ok]
    base = t0.min()
    s, e = (t0 - base) * 0.01, (t1 - base) * 0.01              # microseconds
    dur = e - s
    hw = meta & 0xffffffff
    simd = (hw >> 4) & 3
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    xcc = (meta >> 32) & 0xf
    bodies = (meta >> 36) & 0xfff
    groups = (meta >> 48) & 0xfff
    simd_key = ((((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd).astype(np.int64)
    span = e.max()
    order = np.argsort(e)
    out = {"label": label, "tiles": int(ok.sum()), "span_us": float(span),
           "first_start_spread_us": float(np.quantile(s, 0.99) - s.min()),
           "tile_us_quantiles(50,90,99,max)": [float(np.quantile(dur, q)) for q in (0.5, 0.9, 0.99, 1.0)],
           "done_at_us(50%,90%,99%)": [float(e[order[int(q * len(e)) - 1]]) for q in (0.5, 0.9, 0.99)],
           "shader_MHz_median": float(np.median(cyc / np.maximum(dur, 0.01))),
           "bodies_per_tile(mean,max)": [float(bodies.mean()), int(bodies.max())],
           "groups_per_tile(mean,max)": [float(groups.mean()), int(groups.max())]}
    # waves in flight over time (2 us buckets)
    edges = np.arange(0, span + 2, 2.0)
    inflight = [(int(((s <= a) & (e > a)).sum())) for a in edges]
    out["waves_in_flight_every_2us"] = inflight
    # per-SIMD busy wave-time and last finish
    keys, inv = np.unique(simd_key, return_inverse=True)
    busy = np.bincount(inv, weights=dur)
    last = np.zeros(len(keys))
    np.maximum.at(last, inv, e)
    out["simds_seen"] = int(len(keys))
    out["simd_wave_time_us(median,p90,max)"] = [float(np.median(busy)), float(np.quantile(busy, 0.9)), float(busy.max())]
    out["simd_last_finish_us(median,p10,max)"] = [float(np.median(last)), float(np.quantile(last, 0.1)), float(last.max())]
    out["xcc_tiles"] = np.bincount(xcc.astype(np.int64), minlength=8).tolist()
    # correlation: a tile's duration vs its executed bodies
    out["corr(duration, bodies)"] = float(np.corrcoef(dur, bodies)[0, 1]) if bodies.std() > 0 else None
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--faces", type=int, default=8)
    ap.add_argument("--tune", action="append", default=[])
    ap.add_argument("--out", type=str, default="")
    a = ap.parse_args()
    ver = _lib.load().gcfr_version().decode()
    if "+counters" not in ver:
        raise SystemExit("product build (%s): run with GCFR_HIP_LIB=<counting build>" % ver)
    dev = torch.device("cuda:0")
    B = a.faces
    prm = RenderParams()
    depth, mask, albedo, normals, light, amb = bench.synth_faces(B, 0)
    t = lambda x: torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    args = (t(depth), t(mask), t(light).reshape(B, 1, 3), t(amb).reshape(B, 1), t(normals), t(albedo))
    n_tiles = B * 16 * 64                                      # 16 x 4 tiles of a 256 x 256 face
    saved = {}
    for tune in (a.tune or [""]):
        knobs = {k: int(v) for k, v in (kv.split("=") for kv in tune.split(",") if kv)}
        buf = torch.zeros(_lib.N_COUNTERS + 4 * n_tiles, dtype=torch.int64, device=dev)
        for rep in range(3):                                   # the last repetition is the one analysed (warm caches)
            buf.zero_()
            R.render_fwd(*args, prm, want_argmin=False, options=_lib.options(**knobs, counters=buf.data_ptr()))
            torch.cuda.synchronize()
        rec = buf[_lib.N_COUNTERS:].cpu().numpy().astype(np.uint64).reshape(n_tiles, 4)
        res = analyse(rec.astype(np.int64), tune)
        print(json.dumps(res))
        saved[tune] = rec
    if a.out:
        np.savez_compressed(os.path.join(ROOT, a.out), **{k.replace("=", "_").replace(",", "__"): v for k, v in saved.items()})


if __name__ == "__main__":
    main()
