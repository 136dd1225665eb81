#!/usr/bin/env python3
"""Register / scratch / occupancy table of the march kernels: tools/resusage.py [file.hip] [extra hipcc flags...] [--all]
(hipcc -Rpass-analysis=kernel-resource-usage on csrc/gcfr_march_unit.hip -- the default shape, 16 x 4 tiles and groups of
four, unless -DGCFR_UNIT_TILE_W= / -DGCFR_UNIT_GROUP= say otherwise -- condensed; --all: every kernel of the unit)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 and sys.argv[1].endswith(".hip") else "gcfr_march_unit.hip"
extra = [a for a in sys.argv[1:] if a != "--all" and not a.endswith(".hip")]
if src == "gcfr_march_unit.hip":
    extra += [d for d, key in (("-DGCFR_UNIT_TILE_W=16", "GCFR_UNIT_TILE_W"), ("-DGCFR_UNIT_GROUP=4", "GCFR_UNIT_GROUP"))
              if not any(key in a for a in extra)]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
       "-fno-fast-math", "-munsafe-fp-atomics", "-Rpass-analysis=kernel-resource-usage",
       os.path.join(ROOT, "geomconsistentfr_amd", "csrc", src), "-o", "/tmp/_resusage.so"] + extra
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
    if "error" in line:
        print(line)
    if not m:
        continue
    if m.group(1) == "Function Name":
        cur = {"name": subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
    elif cur is not None:
        cur[m.group(1).split(" ")[0]] = m.group(2)
show_all = "--all" in sys.argv
for r in rows:
    n = r["name"].replace("gcfr::", "").replace("void ", "").split("(")[0]
    if not show_all and "quad" in n and not re.search(r"<\d+, true, (true, |false, )?\d, true", n):
        continue
    print("%-70s sgpr %3s vgpr %3s scratch %4s occ %s lds %s" % (n, r.get("TotalSGPRs"), r.get("VGPRs"), r.get("ScratchSize"),
                                                                 r.get("Occupancy"), r.get("LDS")))
