#!/usr/bin/env python3
"""Instruction census of a march kernel, per stage, from the ISA (CPU; needs hipcc + the ROCm LLVM tools).

    tools/census.py [--kernel 'shadow_fwd_quad_kernel<16, true, 4, true, 0>'] [--out profiles/r05_fixed_cost_census.json] [-DMACRO ...]

How: csrc/gcfr_march_unit.hip is compiled for gfx950 with -gline-tables-only (device only).  The line table gives every
instruction the source line it came from (inlined callees keep THEIR lines: end_point(), unit_normal(), lambert_dot() ...), and the
DWARF inlined-subroutine tree says which instantiation of march_tile() an instruction belongs to -- a grid kernel holds four
(all-ones mask or not) x (bounds variant or rough variant), and a tile executes one (a rough tile: two prologues).  A source line
belongs to the stage named by the last `// census: <stage>` marker above it in its file (`// census: @caller`: the lines below are
helpers that count where they are called from -- the buffer-load wrappers).

The fixed stages (tile set-up, end point, candidate range, bounds set-up, epilogue ...) are straight-line code executed once per
tile, so their static counts ARE the per-tile dynamic counts (both arms of a wave-uniform branch are listed; which one a
workload takes is noted in the output).  Loop stages are static counts per iteration-body as compiled (unrolled bodies count once
per copy) -- their dynamic weight comes from the counters (profiles/pmc_summary.json: VALU per launch minus tiles x fixed).

Checked: the kernel's instruction stream in the -g build equals the product library's (debug info does not change code).
Issue cycles at the guide's SPEC rates: 2 per wave64 f32 / int VALU op, 4 per f64 / cvt / 64-bit int, 8 per f32 transcendental,
16 per f64 transcendental (MI355X_MICROARCH.md; bench.py SPEC_CYCLES)."""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "geomconsistentfr_amd", "csrc")
LLVM = "/opt/rocm/lib/llvm/bin"
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-fno-fast-math", "-munsafe-fp-atomics"]

TRANS = ("v_rcp_", "v_rsq_", "v_sqrt_", "v_exp_", "v_log_", "v_sin_", "v_cos_")


def classify(mn):
    """(unit, class, spec issue cycles) of one mnemonic"""
    if mn.startswith("v_"):
        if mn.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
            return "valu", "lane", 2
        if mn.startswith(TRANS):
            return "valu", "trans_f64" if "_f64" in mn else "trans_f32", 16 if "_f64" in mn else 8
        if mn.startswith("v_cvt_"):
            return "valu", "cvt", 4
        if "_f64" in mn:
            return "valu", "f64", 4
        if re.search(r"_(b64|i64|u64)", mn) and not mn.startswith(("v_cndmask", "v_mov_b64")):
            return "valu", "int64", 4
        if mn.startswith("v_mov_b64"):
            return "valu", "mov64", 2
        if re.search(r"_f32", mn):
            return "valu", "f32", 2
        return "valu", "int32/other", 2
    if mn.startswith(("buffer_", "global_", "flat_", "scratch_")):
        return "vmem", "vmem", 0
    if mn.startswith("ds_"):
        return "lds", "lds", 0
    if mn.startswith("s_load") or mn.startswith("s_buffer_load"):
        return "smem", "smem", 0
    if mn.startswith("s_waitcnt"):
        return "wait", "wait", 0
    if mn.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_barrier", "s_setpc", "s_swappc")):
        return "branch", "branch", 0
    if mn.startswith("s_"):
        return "salu", "salu", 0
    return "other", "other", 0


def stage_maps():
    """{file basename: sorted [(first line, stage)]} from the `// census: <stage>` markers"""
    out = {}
    for f in os.listdir(CSRC):
        marks = []
        for i, line in enumerate(open(os.path.join(CSRC, f), encoding="utf-8"), 1):
            m = re.match(r"\s*// census: (.+?)\s*$", line)
            if m:
                marks.append((i, m.group(1)))
        if marks:
            out[f] = marks
    return out


def stage_of(maps, path, line):
    marks = maps.get(os.path.basename(path))
    if not marks:
        return "(unmarked file: %s)" % os.path.basename(path)
    cur = "(before the first marker: %s)" % os.path.basename(path)
    for first, name in marks:
        if first <= line:
            cur = name
        else:
            break
    return cur


def demangled_kernels(co):
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-t", co], capture_output=True, text=True).stdout.split("\n")
    names = [ln.split()[-1] for ln in dis if " F .text" in ln]
    dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, dem))


def disassemble(co, mangled):
    """[(address, mnemonic, file, line)] of one kernel"""
    dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "-l", co], capture_output=True, text=True,
                         errors="replace").stdout
    out, on, loc = [], False, (None, 0)
    for line in dis.split("\n"):
        m = re.match(r"^[0-9a-f]{8,} <(.+)>:", line)
        if m:
            on = m.group(1) == mangled
            continue
        if not on:
            continue
        m = re.match(r"^; (/.+):(\d+)$", line)
        if m:
            loc = (m.group(1), int(m.group(2)))
            continue
        m = re.match(r"^\s+(\S+).*//\s*([0-9A-Fa-f]+):", line)
        if m and not line.lstrip().startswith(";"):
            out.append((int(m.group(2), 16), m.group(1), loc[0], loc[1]))
    return out


def inlined_tree(co, kernel_dem):
    """Every DW_TAG_inlined_subroutine inside the kernel's subprogram DIE: [(depth, origin name, [(lo, hi)], call file, call line)]
    (depth = nesting level in the dump: the instantiations of march_tile are the outermost ones that carry that name)."""
    txt = subprocess.run([os.path.join(LLVM, "llvm-dwarfdump"), "--debug-info", co], capture_output=True, text=True).stdout
    lines = txt.split("\n")
    want = kernel_dem.replace("void ", "").split("(")[0].replace("gcfr::", "")
    res, i, n = [], 0, len(lines)
    while i < n:
        if "DW_TAG_subprogram" in lines[i]:
            j, name = i + 1, None
            while j < n and lines[j].strip() and "DW_TAG_" not in lines[j]:
                m = re.search(r'DW_AT_name\s+\("(.+)"\)', lines[j])
                if m:
                    name = m.group(1)
                j += 1
            if name == want:
                k = j
                while k < n and "DW_TAG_subprogram" not in lines[k] and "DW_TAG_compile_unit" not in lines[k]:
                    m0 = re.match(r"^0x[0-9a-f]+:(\s+)DW_TAG_inlined_subroutine", lines[k])
                    if m0:
                        depth = len(m0.group(1)) // 2
                        origin, ranges, cfile, cline, lo, q = None, [], None, 0, None, k + 1
                        while q < n and lines[q].strip() and not re.match(r"^0x[0-9a-f]+:", lines[q]):
                            m = re.search(r'DW_AT_abstract_origin\s+\(0x[0-9a-f]+ "(.+)"\)', lines[q])
                            if m:
                                origin = m.group(1)
                            m = re.search(r"\[0x([0-9a-f]+), 0x([0-9a-f]+)\)", lines[q])
                            if m:
                                ranges.append((int(m.group(1), 16), int(m.group(2), 16)))
                            m = re.search(r"DW_AT_low_pc\s+\(0x([0-9a-f]+)\)", lines[q])
                            if m:
                                lo = int(m.group(1), 16)
                            m = re.search(r"DW_AT_high_pc\s+\(0x([0-9a-f]+)\)", lines[q])
                            if m and lo is not None:
                                ranges.append((lo, int(m.group(1), 16)))
                            m = re.search(r'DW_AT_call_file\s+\("(.+)"\)', lines[q])
                            if m:
                                cfile = m.group(1)
                            m = re.search(r"DW_AT_call_line\s+\((\d+)\)", lines[q])
                            if m:
                                cline = int(m.group(1))
                            q += 1
                        res.append((depth, origin or "?", ranges, cfile, cline))
                    k += 1
                return res
        i += 1
    return res


def tile_instantiations(tree):
    """[(name of the march_tile instantiation, [(lo, hi)])]"""
    return [(o, r) for _, o, r, _, _ in tree if o.startswith("march_tile<")]


def describe_instantiation(name):
    """march_tile<TILE_W, EVEN_HALF, WANT_ARGMIN, DEPTH, FUSE_SHADE, SPLIT, ALL_ONES, LDS, OWN, MODE>"""
    a = [x.strip() for x in name[name.index("<") + 1:name.rindex(">")].split(",")]
    mode = {"0": "inline", "1": "bounds variant", "2": "rough variant"}.get(a[9], a[9])
    return ("all-ones mask" if a[6] == "true" else "mask with zeros") + ", " + mode + (", pixels = mask" if a[8] == "true" else "")


def main():
    args = sys.argv[1:]
    kernel = "shadow_fwd_quad_kernel<16, true, 4, true, 0>"
    out_path = None
    defines = []
    product = os.path.join(ROOT, "geomconsistentfr_amd", "lib", "libgcfr_hip.so")
    prebuilt, dump = None, None
    while args:
        a = args.pop(0)
        if a == "--kernel":
            kernel = args.pop(0)
        elif a == "--out":
            out_path = args.pop(0)
        elif a == "--product":
            product = args.pop(0)
        elif a == "--dump":       # print the instructions of the stages whose name contains this string (every instantiation)
            dump = args.pop(0)
        elif a == "--co":         # a code object compiled earlier with -gline-tables-only (skips the compilation)
            prebuilt = args.pop(0)
        elif a == "--csrc":       # another copy of csrc/ (an earlier commit's, for a like-for-like table)
            global CSRC
            CSRC = os.path.abspath(args.pop(0))
        else:
            defines.append(a)
    tw = next((d.split("=")[1] for d in defines if d.startswith("-DGCFR_UNIT_TILE_W=")), "16")
    gr = next((d.split("=")[1] for d in defines if d.startswith("-DGCFR_UNIT_GROUP=")), "4")
    defines = [d for d in defines if not d.startswith(("-DGCFR_UNIT_TILE_W=", "-DGCFR_UNIT_GROUP="))]
    with tempfile.TemporaryDirectory(prefix="gcfr_census_") as tmp:
        obj, co = os.path.join(tmp, "march_g.o"), os.path.join(tmp, "march_g.co")
        cmd = ["/opt/rocm/bin/hipcc"] + FLAGS + ["--cuda-device-only", "-gline-tables-only", "-DGCFR_UNIT_TILE_W=" + tw,
                                                  "-DGCFR_UNIT_GROUP=" + gr] + defines + ["-c", os.path.join(CSRC, "gcfr_march_unit.hip"), "-o", obj]
        if prebuilt:
            co = prebuilt
        else:
            subprocess.run(cmd, check=True)
            subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + obj,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], check=True)
        kernels = demangled_kernels(co)
        hits = [m for m, d in kernels.items() if ("gcfr::" + kernel + "(") in d or d.startswith("void gcfr::" + kernel + "(")]
        if len(hits) != 1:
            sys.exit("kernel %r: %d matches among %s" % (kernel, len(hits), sorted(kernels.values())[:6]))
        mangled = hits[0]
        ins = disassemble(co, mangled)
        tree = inlined_tree(co, kernels[mangled])
        tiles = tile_instantiations(tree)
        # the product library's stream of the same kernel (debug info must not have changed the code)
        same_as_product = None
        if os.path.exists(product) and not defines:
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import compare_device_code as cdc
            with tempfile.TemporaryDirectory() as t2:
                prod = cdc.kernels_of(product, t2)
            stream = [v for (_, sym), v in prod.items() if sym == mangled]
            mine = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, errors="replace").stdout
            cur, got = None, []
            for line in mine.split("\n"):
                m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                if m:
                    cur = m.group(1)
                elif cur == mangled:
                    t = line.split("//")[0].strip()
                    if t:
                        got.append(t)
            same_as_product = bool(stream) and stream[0] == got
    maps = stage_maps()

    def inst_of(addr):
        for name, rs in tiles:
            for lo, hi in rs:
                if lo <= addr < hi:
                    return name
        return None

    def caller_in_sources(addr, f, ln):
        """an instruction whose own line lies in a header outside csrc/ (fminf, floor, expf, __mul24 ...): the line of the innermost
        call site that does lie in csrc/, through the inlined-subroutine chain"""
        def ours(ff, ll):  # a line of csrc/ whose stage is not "@caller" (tiny helpers -- buffer loads -- count where they are called)
            return bool(ff) and os.path.basename(ff) in maps and stage_of(maps, ff, ll) != "@caller"
        if ours(f, ln):
            return f, ln
        chain = sorted((d for d in tree if any(lo <= addr < hi for lo, hi in d[2])), key=lambda d: -d[0])
        for _, _, _, cfile, cline in chain:
            if ours(cfile, cline):
                return cfile, cline
        return f, ln

    table = collections.OrderedDict()
    for addr, mn, f, ln in ins:
        f, ln = caller_in_sources(addr, f, ln)
        inst = inst_of(addr)
        where = describe_instantiation(inst) if inst else "kernel level (march_grid, image statistics)"
        stage = stage_of(maps, f, ln) if f else "(no line)"
        unit, cls, cyc = classify(mn)
        if dump and dump in stage:
            print("%-34s %-22s %s:%d  [%s]" % (where[:34], mn, os.path.basename(f or "?"), ln, stage[:40]))
        e = table.setdefault(where, collections.OrderedDict()).setdefault(stage, collections.Counter())
        e[unit] += 1
        if unit == "valu":
            e["valu:" + cls] += 1
            e["spec_issue_cycles"] += cyc
    result = {"kernel": kernel, "defines": defines, "tile_w": int(tw), "group": int(gr), "instructions": len(ins),
              "same_instruction_stream_as_product_library": same_as_product,
              "how": "static per-stage counts from the ISA of a -gline-tables-only build: line table -> source line -> `// census:` marker; "
                     "DWARF inlined-subroutine ranges -> march_tile instantiation (tools/census.py)",
              "by_instantiation": {w: {s: dict(c) for s, c in st.items()} for w, st in table.items()}}
    # summary: fixed (non-loop) stages of the instantiation the bench faces run
    for w, st in table.items():
        fixed = {s: c for s, c in st.items() if not s.startswith("loop:")}
        result.setdefault("fixed_valu_per_tile", {})[w] = sum(c["valu"] for c in fixed.values())
        result.setdefault("fixed_spec_issue_cycles_per_tile", {})[w] = sum(c["spec_issue_cycles"] for c in fixed.values())
    txt = json.dumps(result, indent=1)
    if out_path:
        with open(out_path, "w") as f:
            f.write(txt + "\n")
    for w, st in table.items():
        print("== %s" % w)
        print("   %-78s %5s %5s %5s %5s %5s %5s %6s" % ("stage", "VALU", "f64", "trans", "cvt", "SALU", "VMEM", "cycles"))
        for s, c in st.items():
            print("   %-78s %5d %5d %5d %5d %5d %5d %6d" % (s[:78], c["valu"], c["valu:f64"] + c["valu:int64"], c["valu:trans_f32"] + c["valu:trans_f64"],
                                                           c["valu:cvt"], c["salu"], c["vmem"], c["spec_issue_cycles"]))
        print("   %-78s %5d %37s %6d" % ("fixed stages (everything but `loop:`)", result["fixed_valu_per_tile"][w], "",
                                         result["fixed_spec_issue_cycles_per_tile"][w]))
    print("same instruction stream as the product library:", same_as_product)


if __name__ == "__main__":
    main()
