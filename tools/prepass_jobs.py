#!/usr/bin/env python3
"""Which job of the prepass kernel sets its duration?  (GPU)

    tools/prepass_jobs.py build            (CPU)  lib/prepass_jobs_<mask>.so for the masks below: -DGCFR_FAST_BUILD -DGCFR_PREPASS_JOBS=<mask>
    tools/prepass_jobs.py run [--faces 8]  (GPU)  per library: the prepass alone (gcfr_options.phase = 1) as a hipGraph, replayed one at a
                                                  time on one stream; fenced wall time per replay and the HIP-event time of 200 single
                                                  replays -> gpurun_out/prepass_jobs.json

build_quad_kernel's grid is [horizon tables | depth-bounds tiles | statistics chunks | (bitmap) | repack]; -DGCFR_PREPASS_JOBS leaves the
grid as it is and makes the blocks of the jobs whose bit is clear return at once (1 horizon, 2 bounds, 4 statistics, 8 bitmap, 16 repack),
so a single job's time includes the dispatch of the other jobs' empty blocks -- which is what it would cost next to faster neighbours.
The workspace such a library leaves is garbage: only the prepass is timed."""
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_DIR = os.path.join(REPO, "geomconsistentfr_amd", "lib")
MASKS = [("all", 31), ("none", 0), ("horizon", 1), ("bounds", 2), ("statistics", 4), ("repack", 16), ("all but horizon", 30),
         ("all but repack", 15), ("all but bounds", 29), ("all but statistics", 27)]


def lib_of(mask):
    return os.path.join(LIB_DIR, "prepass_jobs_%d.so" % mask)


def build():
    sys.path.insert(0, REPO)
    from geomconsistentfr_amd import build as b
    for _, m in MASKS:
        t = time.time()
        b.compile_and_link(lib_of(m), defines=["-DGCFR_FAST_BUILD", "-DGCFR_PREPASS_JOBS=%d" % m], jobs=8)
        print("prepass_jobs_%d.so  %.0f s" % (m, time.time() - t), flush=True)


def one(faces):
    """(in a subprocess with GCFR_HIP_LIB set)"""
    sys.path.insert(0, REPO)
    import numpy as np
    import torch
    import bench

    class One:  # a minimal stand-in for bench.Ranks: one rank, fences = synchronize
        dev, rank, world = torch.device("cuda:0"), 0, 1

        def fence(self):
            torch.cuda.synchronize()

        def max_over_ranks(self, s):
            return s
    rig = bench.RenderRig(One(), faces, streams=1, from_depth=True)
    p0 = rig.plans[0]
    p0.capture_split(*rig.inputs[0])
    st = rig.streams[0]
    with torch.cuda.stream(st):
        for _ in range(200):
            p0.replay_prepass()
    torch.cuda.synchronize()
    walls = []
    for _ in range(7):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with torch.cuda.stream(st):
            for _ in range(2000):
                p0.replay_prepass()
        torch.cuda.synchronize()
        walls.append((time.perf_counter() - t0) / 2000)
    evs = []
    with torch.cuda.stream(st):
        for _ in range(200):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            p0.replay_prepass()
            e1.record(st)
            evs.append((e0, e1))
    torch.cuda.synchronize()
    ev_ms = sorted(e0.elapsed_time(e1) for e0, e1 in evs)
    print(json.dumps({"wall_us_per_replay_median": 1e6 * float(np.median(walls)), "wall_us_per_replay_min": 1e6 * min(walls),
                      "event_us_median": 1e3 * ev_ms[len(ev_ms) // 2], "event_us_min": 1e3 * ev_ms[0]}))


def run(faces):
    out = {"faces": faces, "how": __doc__.split("\n\n")[1], "libraries": {}}
    for name, m in MASKS:
        lib = lib_of(m)
        if not os.path.exists(lib):
            continue
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "one", str(faces)], env=dict(os.environ, GCFR_HIP_LIB=lib),
                           capture_output=True, text=True)
        try:
            out["libraries"][name] = dict(json.loads(r.stdout.strip().splitlines()[-1]), mask=m)
        except Exception:
            out["libraries"][name] = {"mask": m, "error": (r.stderr or r.stdout)[-600:]}
        print(name, out["libraries"][name], flush=True)
    os.makedirs(os.path.join(REPO, "gpurun_out"), exist_ok=True)
    with open(os.path.join(REPO, "gpurun_out", "prepass_jobs.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else ""
    if what == "build":
        build()
    elif what == "one":
        one(int(sys.argv[2]))
    elif what == "run":
        run(int(sys.argv[sys.argv.index("--faces") + 1]) if "--faces" in sys.argv else 8)
    else:
        sys.exit(__doc__)
