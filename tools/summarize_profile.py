#!/usr/bin/env python3
"""Condense a tools/prof.sh run (gpurun_out/prof_<tag>/) into the small, committed files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_pmc.json           per-kernel mean of every PMC counter collected (separate passes)
  profiles/pmc_summary.json         what bench.py reports as roofline.traffic (HBM bytes per shadow launch)

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
tallies 128-B requests at 64 B, so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main(tag):
    src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, tag + "_kernel_stats.csv"))
    per_kernel = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per_kernel[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    summary = {k: {c: {"mean": sum(v) / len(v), "n": len(v)} for c, v in cs.items()} for k, cs in per_kernel.items()}
    stats = {r["Name"].split("(")[0].replace("void ", ""): r
             for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv")))}
    for k in summary:
        if k in stats:
            summary[k]["_avg_ns"] = float(stats[k]["AverageNs"])
            summary[k]["_calls"] = int(stats[k]["Calls"])
    json.dump(summary, open(os.path.join(dst, tag + "_pmc.json"), "w"), indent=1, sort_keys=True)
    shadow = next((k for k in summary if "shadow_fwd" in k or "render_fwd" in k), None)
    if shadow and "FETCH_SIZE" in summary[shadow]:
        fetch_kib = summary[shadow]["FETCH_SIZE"]["mean"]
        write_kib = summary[shadow].get("WRITE_SIZE", {"mean": 0.0})["mean"]
        out = {"tag": tag, "kernel": shadow, "FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
               "correction": "read side x2 on gfx950 (MI355X_MICROARCH.md HBM)",
               "shadow_fwd_hbm_bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0,
               "avg_launch_ns": summary[shadow].get("_avg_ns"),
               "valu_insts_per_launch": summary[shadow].get("SQ_INSTS_VALU", {}).get("mean")}
        json.dump(out, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
        print(json.dumps(out, indent=1))
    for k, cs in summary.items():
        print(k, {c: (round(v["mean"], 1) if isinstance(v, dict) else v) for c, v in cs.items()})


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r01")
