"""Builds libgcfr_hip.so (hand-written HIP kernels + C ABI, include/gcfr.h) for gfx950, in-tree.

hipcc cross-compiles without a GPU; the .so is git-ignored but travels to the GPU box with the
snapshot.  -ffp-contract=off is part of the numerical contract (see csrc/gcfr_device.hpp).

The march kernels are templates over (tile width, samples per group); each shape is its own translation unit
(csrc/gcfr_march_unit.hip compiled with -DGCFR_UNIT_TILE_W / -DGCFR_UNIT_GROUP) and the units compile in parallel:
~95 s as one translation unit, ~20 s on 8 cores this way.

    python geomconsistentfr_amd/build.py                          # the product library
    python geomconsistentfr_amd/build.py --variant NAME -DMACRO   # lib/NAME.so (tools/build_variant.sh), select with GCFR_HIP_LIB
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys
import tempfile

PKG = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG, "csrc")
LIB_DIR = os.path.join(PKG, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libgcfr_hip.so")
SOURCES = ["gcfr_shadow.hip", "gcfr_shade.hip", "gcfr_backward.hip", "gcfr_normals.hip", "gcfr_postprocess.hip",
           "gcfr_dataset.hip"]
MARCH_UNIT = "gcfr_march_unit.hip"
# (tile width, samples per group): the default shape first -- it is the largest unit (it also holds the LDS-staged kernels)
MARCH_UNITS = [(16, 4), (16, 2), (16, 1), (8, 4), (8, 2), (8, 1), (32, 4), (32, 2), (32, 1), (64, 4), (64, 2), (64, 1)]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-fno-fast-math", "-munsafe-fp-atomics", "-Wall"]


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libgcfr_hip.so cannot be built")


HASH_PATH = os.path.join(LIB_DIR, "libgcfr_hip.srchash")


def _code_only(text: str) -> str:
    """C / C++ source without comments and without blank lines or trailing blanks: what the compiler sees.  The content hash is
    taken over this, so that editing a comment neither forces a rebuild nor marks the committed profiles as stale."""
    out, i, n = [], 0, len(text)
    while i < n:
        c = text[i]
        if c == "'" and i > 0 and i + 1 < n and text[i - 1].isalnum() and text[i + 1].isalnum() and _in_number(text, i):
            out.append(c)                             # C++14 digit separator (1'000'000), not a character literal
            i += 1
        elif c == '"' and _prefix_token(text, i) in ("R", "u8R", "uR", "UR", "LR"):
            k = text.find("(", i)                     # raw string R"delim( ... )delim": copied verbatim, nothing inside is a comment
            end = -1 if k < 0 else text.find(")" + text[i + 1:k] + '"', k)
            j = n if end < 0 else end + (k - i) + 1
            out.append(text[i:j])
            i = j
        elif c == '"' or c == "'":                    # string / character literal: copied verbatim
            j = i + 1
            while j < n and text[j] != c:
                j += 2 if text[j] == "\\" else 1
            out.append(text[i:j + 1])
            i = j + 1
        elif text.startswith("//", i):
            j = text.find("\n", i)
            i = n if j < 0 else j
        elif text.startswith("/*", i):
            j = text.find("*/", i + 2)
            out.append(" ")
            i = n if j < 0 else j + 2
        else:
            out.append(c)
            i += 1
    lines = [ln.rstrip() for ln in "".join(out).split("\n")]
    return "\n".join(ln for ln in lines if ln.strip())


def _prefix_token(text: str, i: int) -> str:
    """the identifier characters immediately in front of text[i] (a string literal's encoding / raw prefix, if any)"""
    j = i
    while j > 0 and (text[j - 1].isalnum() or text[j - 1] == "_"):
        j -= 1
    return text[j:i]


def _in_number(text: str, i: int) -> bool:
    """Is the apostrophe at text[i] inside a numeric literal?  Walk back over [0-9a-zA-Z'.]: a pp-number starts with a digit
    (or a dot followed by one)."""
    j = i
    while j > 0 and (text[j - 1].isalnum() or text[j - 1] in "'._"):
        j -= 1
    return j < i and (text[j].isdigit() or (text[j] == "." and j + 1 < i and text[j + 1].isdigit()))


def source_hash() -> str:
    """sha256 over the flags, the list of march units and the CODE (comments stripped) of every file the library is compiled
    from (csrc/*, include/gcfr.h)."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + " " + repr(MARCH_UNITS)).encode())
    deps = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(PKG, "..", "include", "gcfr.h")]
    for d in deps:
        h.update(os.path.basename(d).encode())
        with open(d, "r", encoding="utf-8") as f:
            h.update(_code_only(f.read()).encode())
    return h.hexdigest()


def full_bytes_hash() -> str:
    """sha256 over the raw bytes of the same files: the fallback beside the code-only hash (second line of the .srchash
    file).  `GCFR_STRICT_HASH=1` makes needs_build() compare it too, so that a change the comment stripper mis-reads as
    "comment only" cannot leave a stale library behind where that matters (CI; build(force=True) always rebuilds anyway)."""
    import hashlib
    h = hashlib.sha256((" ".join(FLAGS) + " " + repr(MARCH_UNITS)).encode())
    for d in sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC)) + [os.path.join(PKG, "..", "include", "gcfr.h")]:
        h.update(os.path.basename(d).encode())
        with open(d, "rb") as f:
            h.update(f.read())
    return h.hexdigest()


def recorded_hashes():
    """(code-only hash, full-bytes hash | None) recorded beside the library, or (None, None)."""
    try:
        with open(HASH_PATH) as f:
            lines = f.read().split()
    except OSError:
        return None, None
    return (lines[0] if lines else None), (lines[1] if len(lines) > 1 else None)


def needs_build() -> bool:
    """True when the library is missing or was built from other sources (content hash recorded beside the .so --
    not mtimes, which a snapshot copy to the GPU box does not preserve)."""
    if not os.path.exists(LIB_PATH) or not os.path.exists(HASH_PATH):
        return True
    code, full = recorded_hashes()
    if code != source_hash():
        return True
    return os.environ.get("GCFR_STRICT_HASH") == "1" and full is not None and full != full_bytes_hash()


def compile_and_link(out: str, defines=(), verbose: bool = False, jobs=None) -> str:
    """Every translation unit to an object (in parallel), then one link.  -DGCFR_FAST_BUILD: the default march shape only."""
    hipcc = _hipcc()
    defines = list(defines)
    hashes = source_hash() + "\n" + full_bytes_hash() + "\n" + " ".join(defines) + "\n"   # (of the sources as they are NOW)
    cflags = [f for f in FLAGS if f != "-shared"] + defines
    units = MARCH_UNITS[:1] if "-DGCFR_FAST_BUILD" in defines else MARCH_UNITS
    os.makedirs(os.path.dirname(out), exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="gcfr_build_") as tmp:
        jobs_list = [([hipcc] + cflags + ["-DGCFR_UNIT_TILE_W=%d" % tw, "-DGCFR_UNIT_GROUP=%d" % g, "-c",
                      os.path.join(CSRC, MARCH_UNIT), "-o", os.path.join(tmp, "march_%d_%d.o" % (tw, g))]) for tw, g in units]
        jobs_list += [([hipcc] + cflags + ["-c", os.path.join(CSRC, s), "-o", os.path.join(tmp, s.replace(".hip", ".o"))])
                      for s in SOURCES]

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed (%d): %s\n%s" % (r.returncode, " ".join(cmd), r.stderr[-4000:]))
            if r.stderr.strip():
                sys.stderr.write(r.stderr)
            return cmd[-1]

        with concurrent.futures.ThreadPoolExecutor(max_workers=jobs or min(len(jobs_list), os.cpu_count() or 4)) as ex:
            objs = list(ex.map(run, jobs_list))
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
    # beside every build: what it was built from (line 1 the code-only hash, 2 the raw-bytes hash, 3 the -D flags); tests that run a
    # variant (tests/test_gpu_audit.py) refuse one that no longer matches the sources
    with open(os.path.splitext(out)[0] + ".srchash", "w") as f:
        f.write(hashes)
    return out


def variant_is_current(path: str) -> bool:
    """a variant library built by compile_and_link() from the sources as they are now (its .srchash beside it says so)"""
    try:
        with open(os.path.splitext(path)[0] + ".srchash") as f:
            return f.read().split("\n")[0] == source_hash()
    except OSError:
        return False


def build(force: bool = False, verbose: bool = False) -> str:
    if not force and not needs_build():
        return LIB_PATH
    compile_and_link(LIB_PATH, verbose=verbose)
    with open(HASH_PATH, "w") as f:
        f.write(source_hash() + "\n" + full_bytes_hash() + "\n")
    return LIB_PATH


if __name__ == "__main__":
    argv = sys.argv[1:]
    if argv and argv[0] == "--variant":
        print(compile_and_link(os.path.join(LIB_DIR, argv[1] + ".so"), defines=argv[2:], verbose=True))
    else:
        print(build(force=True, verbose=True))
