#!/usr/bin/env python3
"""Which kernels of two builds of the library differ?  (CPU; needs the ROCm LLVM tools.)
usage: tools/compare_device_code.py OLD.so NEW.so
Extracts the gfx950 code objects of both, disassembles them and compares every kernel's instruction stream (mnemonics and
operands, addresses and encodings dropped).  Used when a source change should leave most kernels alone -- e.g. before keeping
PMC profiles that were collected on the previous build: profiles/r04_report.md."""
import collections
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(so, tmp):
    d = os.path.join(tmp, os.path.basename(so) + ".d")
    os.makedirs(d)
    shutil.copy(so, os.path.join(d, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, capture_output=True, check=True)
    out = collections.OrderedDict()
    for co in sorted(glob.glob(os.path.join(d, "lib.so.*gfx950"))):
        dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", co], capture_output=True, text=True, errors="replace").stdout
        cur = None
        for line in dis.split("\n"):
            m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
            if m:
                cur = (os.path.basename(co).split(".")[2], m.group(1))
                out[cur] = []
            elif cur is not None:
                ins = line.split("//")[0].strip()
                if ins:
                    out[cur].append(ins)
    return out


def main(old, new):
    with tempfile.TemporaryDirectory() as tmp:
        a, b = kernels_of(old, tmp), kernels_of(new, tmp)
    same = [k for k in a if k in b and a[k] == b[k]]
    changed = [k for k in a if k in b and a[k] != b[k]]
    only = [k for k in a if k not in b] + [k for k in b if k not in a]
    names = collections.Counter()
    for _, sym in changed + only:
        dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        names[re.sub(r"<.*", "", dem.replace("void ", ""))] += 1
    print("%d kernels identical, %d changed, %d only in one build" % (len(same), len(changed), len(only)))
    for n, c in sorted(names.items()):
        print("  %-60s %d instantiation(s)" % (n, c))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
