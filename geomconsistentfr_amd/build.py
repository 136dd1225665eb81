"""Builds libgcfr_hip.so (hand-written HIP kernels + C ABI, include/gcfr.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
snapshot.  -ffp-contract=off is part of the numerical contract (see csrc/gcfr_device.hpp).
"""
import os
import shutil
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgcfr_hip.so")
SOURCES = ["gcfr_shadow.hip", "gcfr_shade.hip", "gcfr_backward.hip", "gcfr_normals.hip", "gcfr_postprocess.hip",
           "gcfr_dataset.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-fast-math", "-munsafe-fp-atomics", "-Wall"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgcfr_hip.so cannot be built")


HASH_PATH = os.path.join(LIB_DIR, "libgcfr_hip.srchash")


def source_hash() -> str:
    """sha256 over the flags and every file the library is compiled from (csrc/*, include/gcfr.h)."""
    import hashlib
    h = hashlib.sha256(" ".join(FLAGS).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(PKG, "..", "include", "gcfr.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def needs_build() -> bool:
    """True when the library is missing or was built from other sources (content hash recorded beside the .so --
    not mtimes, which a snapshot copy to the GPU box does not preserve)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    with open(HASH_PATH) as f:
        return f.read().strip() != source_hash()


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    os.makedirs(LIB_DIR, exist_ok=True)
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    with open(HASH_PATH, "w") as f:
        f.write(source_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    print(build(force=True, verbose=True))
