#!/usr/bin/env python3
"""Condense tools/prof.sh runs (gpurun_out/prof_<tag>_fwd, prof_<tag>_bwd) into the small, committed files under profiles/:

  profiles/<tag>_{fwd,bwd}_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<tag>_{fwd,bwd}_pmc.json           per-kernel mean of every PMC counter collected (separate --pmc passes)
  profiles/<tag>_valu_cost_table.json         issue cost per VALU instruction class, from tools/ubench_valu (profiles/<tag>_ubench_valu.txt)
  profiles/pmc_summary.json                   what bench.py reads: per dominant kernel, HBM bytes per launch, VALU
                                              instructions per launch by class, and the VALU issue time that mix demands

HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B
requests at 64 B, so the read side is doubled.

VALU roofline.  SQ_INSTS_VALU_* split a kernel's VALU wave-instructions by class; tools/ubench_valu measures what one
SIMD sustains for a pure stream of each class (8 waves/SIMD, 16 independent chains, ~1 M instructions per wave, wall
clock at the nominal 2.4 GHz -- i.e. DVFS-inclusive).  issue_cycles_per_launch = sum_class n_class * cost_class is the
SIMD time the kernel's instruction mix needs at those rates; against 1024 SIMDs * 2.4 GHz * launch duration it is the
fraction of the chip's VALU issue capacity the kernel uses -- <= 1 by construction, and the honest "how close to a
hardware limit" for a kernel whose gathers are cache-served.
usage: tools/summarize_profile.py <tag>      (expects gpurun_out/prof_<tag>_fwd, optionally prof_<tag>_bwd, prof_<tag>_fwd128,
       profiles/<tag>_ubench_valu.txt, profiles/<tag>_work_counts.json)
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_SIMD, NOMINAL_HZ = 1024, 2.4e9

# PMC class -> ubench row(s) whose sustained cost stands for it (mean when several)
CLASS_ROWS = {
    "ADD_F32": ["v_add_f32"], "MUL_F32": ["v_mul_f32"], "FMA_F32": ["v_fma_f32"],
    "ADD_F64": ["v_add_f64"], "MUL_F64": ["v_mul_f64"], "FMA_F64": ["v_fma_f64"],
    "CVT": ["v_cvt_i32_f64", "v_cvt_f64_i32", "v_cvt_f32_f64", "v_cvt_f64_f32", "v_cvt_i32_f32"],
    "INT32": ["v_add_u32", "v_lshl_add_u32", "v_mul_i32_i24"], "INT64": ["v_lshl_add_u32", "v_mul_i32_i24"],
}


def cost_table(tag):
    """{class: nominal-2.4-GHz cycles per wave64 instruction per SIMD}, from the w=8 wall column of the ubench."""
    path = os.path.join(ROOT, "profiles", tag + "_ubench_valu.txt")
    if not os.path.exists(path):      # the per-class issue costs are a property of the chip: round 2's table serves later rounds
        path = os.path.join(ROOT, "profiles", "r02_ubench_valu.txt")
    rows = {}
    for line in open(path):
        m = re.match(r"(\S.*?)\s+w=1:", line)
        w8 = re.search(r"w=8:\s*[\d.]+ \(wall\s*([\d.]+)", line)
        if m and w8:
            rows[m.group(1).strip()] = float(w8.group(1))
    cost = {c: sum(rows[r] for r in rs) / len(rs) for c, rs in CLASS_ROWS.items()}
    # transcendentals (v_rcp / v_sqrt / v_exp ...) issue at quarter rate; everything unclassified (v_mov, v_cndmask,
    # v_cmp, min / max, bit ops, DPP, readlane) is priced between the 2.7-cycle f32 ops and the 4.3-cycle VOP3 ops
    cost["TRANS_F32"] = 4.0 * cost["ADD_F32"]
    cost["TRANS_F64"] = 4.0 * cost["ADD_F64"]
    cost["OTHER"] = 0.5 * (cost["ADD_F32"] + rows["v_min_f32"])
    return cost, rows


def kernel_means(src):
    per = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in sorted(glob.glob(os.path.join(src, "pmc_*", "pmc_counter_collection.csv"))):
        for row in csv.DictReader(open(f)):
            k = row["Kernel_Name"].split("(")[0].replace("void ", "")
            per[k][row["Counter_Name"]].append(float(row["Counter_Value"]))
    out = {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in per.items()}
    stats = {r["Name"].split("(")[0].replace("void ", ""): r
             for r in csv.DictReader(open(os.path.join(src, "trace", "trace_kernel_stats.csv")))}
    for k in out:
        if k in stats:
            out[k]["_avg_ns"] = float(stats[k]["AverageNs"])
            out[k]["_calls"] = int(stats[k]["Calls"])
    return out


def valu_demand(c, cost):
    classes = {k: c.get("SQ_INSTS_VALU_" + k, 0.0) for k in cost if k != "OTHER"}
    total = c["SQ_INSTS_VALU"]
    classes["OTHER"] = max(total - sum(classes.values()), 0.0)
    cycles = sum(classes[k] * cost[k] for k in classes)
    return {"insts_per_launch": total, "by_class": classes, "issue_cycles_per_launch": cycles,
            "mean_issue_cycles_per_inst": cycles / total}


def kernel_entry(name, c, cost):
    fetch_kib, write_kib = c.get("FETCH_SIZE", 0.0), c.get("WRITE_SIZE", 0.0)
    e = {"kernel": name, "avg_launch_ns_under_rocprofv3": c.get("_avg_ns"),
         "hbm": {"FETCH_SIZE_KiB_raw": fetch_kib, "WRITE_SIZE_KiB_raw": write_kib,
                 "correction": "read side x2 on gfx950 (MI355X_MICROARCH.md HBM)",
                 "bytes_per_launch": (2.0 * fetch_kib + write_kib) * 1024.0},
         "valu": valu_demand(c, cost),
         "waves": c.get("SQ_WAVES"), "wave_cycles_quad": c.get("SQ_WAVE_CYCLES"), "wait_any_quad": c.get("SQ_WAIT_ANY"),
         "l1_accesses": c.get("TCP_TOTAL_CACHE_ACCESSES_sum"), "l2_requests": c.get("TCC_REQ_sum"),
         "l2_hits": c.get("TCC_HIT_sum")}
    if c.get("_avg_ns"):
        e["valu"]["frac_of_issue_capacity_under_rocprofv3"] = e["valu"]["issue_cycles_per_launch"] / (
            N_SIMD * NOMINAL_HZ * c["_avg_ns"] * 1e-9)
    return e


def main(tag):
    dst = os.path.join(ROOT, "profiles")
    cost, rows = cost_table(tag)
    json.dump({"source": "profiles/%s_ubench_valu.txt (or round 2's when absent), column w=8 'wall' (nominal 2.4 GHz cycles per wave64 instruction per SIMD)" % tag,
               "per_class": cost, "ubench_rows": rows}, open(os.path.join(dst, tag + "_valu_cost_table.json"), "w"), indent=1)
    # the instruction counts / traffic below belong to ONE build of the library: bench.py compares this hash with the loaded
    # library's and marks its roofline `stale` when they differ
    try:
        lib_hash = open(os.path.join(ROOT, "geomconsistentfr_amd", "lib", "libgcfr_hip.srchash")).read().split()[0]
    except OSError:
        lib_hash = None
    summary = {"tag": tag, "library_srchash": lib_hash, "n_simd": N_SIMD, "nominal_hz": NOMINAL_HZ, "kernels": {}}
    # fwd128: the same march on a launch of 128 faces (tools/prof.sh <tag>_fwd128 fwd --faces 128) -- a launch long enough that its
    # tail does not matter: the instruction count behind bench.py's `roofline.saturated`
    for leg, pick in (("fwd", "shadow_fwd_quad"), ("bwd", "render_bwd_single_light"), ("fwd128", "shadow_fwd_quad")):
        src = os.path.join(ROOT, "gpurun_out", "prof_%s_%s" % (tag, leg))
        if not os.path.isdir(src):
            continue
        shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, "%s_%s_kernel_stats.csv" % (tag, leg)))
        means = kernel_means(src)
        json.dump(means, open(os.path.join(dst, "%s_%s_pmc.json" % (tag, leg)), "w"), indent=1, sort_keys=True)
        for k, c in means.items():
            if pick in k and "SQ_INSTS_VALU" in c:
                summary["kernels"]["fwd_b128" if leg == "fwd128" else leg] = kernel_entry(k, c, cost)
                if leg == "fwd128":
                    summary["kernels"]["fwd_b128"]["faces_per_launch"] = 128
            if leg == "bwd" and "shadow_fwd_quad_argmin" in k and "SQ_INSTS_VALU" in c:
                summary["kernels"]["fwd_training_march"] = kernel_entry(k, c, cost)
    wc = os.path.join(dst, tag + "_work_counts.json")
    if os.path.exists(wc) and "fwd" in summary["kernels"]:
        w = json.load(open(wc))
        f = summary["kernels"]["fwd"]
        f["work"] = {"nominal_ray_steps": w["nominal_ray_steps"], "executed_ray_steps_wave_level": w["executed_ray_steps_wave_level"],
                     "executed_fraction_of_nominal": w["executed_fraction_of_nominal"], "counters": w["counters"],
                     "valu_per_executed_ray_step": f["valu"]["insts_per_launch"] / max(w["executed_ray_steps_wave_level"] / 64.0, 1.0) / 64.0 * 64.0 / 64.0,
                     "valu_wave_insts_per_executed_wave_step": f["valu"]["insts_per_launch"] / max(w["executed_ray_steps_wave_level"] / 64.0, 1.0),
                     "valu_wave_insts_per_nominal_wave_step": f["valu"]["insts_per_launch"] / (w["nominal_ray_steps"] / 64.0)}
        f["work"].pop("valu_per_executed_ray_step")
        # L1: bytes the vector cache served per second of kernel time vs its peak (64 B / clk / CU)
        if f.get("l1_accesses") and f.get("avg_launch_ns_under_rocprofv3"):
            f["l1"] = {"accesses_per_launch": f["l1_accesses"], "note": "TCP_TOTAL_CACHE_ACCESSES: 64-B granules",
                       "bytes_per_s": f["l1_accesses"] * 64.0 / (f["avg_launch_ns_under_rocprofv3"] * 1e-9),
                       "peak_bytes_per_s": 256 * 64.0 * NOMINAL_HZ,
                       "frac": f["l1_accesses"] * 64.0 / (f["avg_launch_ns_under_rocprofv3"] * 1e-9) / (256 * 64.0 * NOMINAL_HZ)}
    # keys bench.py has read since round 1 (kept for the forward headline kernel)
    if "fwd" in summary["kernels"]:
        f = summary["kernels"]["fwd"]
        summary["shadow_fwd_hbm_bytes_per_launch"] = f["hbm"]["bytes_per_launch"]
        summary["valu_insts_per_launch"] = f["valu"]["insts_per_launch"]
    json.dump(summary, open(os.path.join(dst, "pmc_summary.json"), "w"), indent=1)
    for leg, e in summary["kernels"].items():
        v = e["valu"]
        print("%-20s %-58s %.1f us  VALU %.2f M inst  %.2f cyc/inst  frac %.3f  HBM %.1f MB" % (
            leg, e["kernel"][-58:], (e["avg_launch_ns_under_rocprofv3"] or 0) / 1e3, v["insts_per_launch"] / 1e6,
            v["mean_issue_cycles_per_inst"], v.get("frac_of_issue_capacity_under_rocprofv3", float("nan")),
            e["hbm"]["bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r02")
