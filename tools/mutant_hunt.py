#!/usr/bin/env python3
"""Search for inputs on which a mutant of the march (csrc/gcfr_mutants.hpp) gives other bits than the unmutated build (GPU).

    tools/mutant_hunt.py [--mutants 2,3,...] [--families A,B,...] [--seeds N] [--out gpurun_out/hunt.json]

The scene FAMILIES (tests/margin_scenes.py) are built from the geometry of the mechanisms the mutants break -- rays running parallel
to a planar surface a hair above it (every sample a near-tie of the running minimum: the bounds' error terms decide), grazing
and overhead lights, distances around the masked value 1e6, walls one cell behind a ray, diamond masks against diagonal rays,
sample tables at the edge of what the prepass accepts ... -- with their free parameters drawn from a seed.  Every scene goes
through lib/mut_0.so (the fast build without a mutation) and through each mutant's library, same inputs, same options; a
difference in min_dist or argmin is a kill, recorded with (family, seed).  The killing scenes become the directed tests of
tests/test_gpu_margins.py, where they are compared with the C oracle."""
import ctypes
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
LIB_DIR = os.path.join(ROOT, "geomconsistentfr_amd", "lib")
_p, _i, _f = ctypes.c_void_p, ctypes.c_int32, ctypes.c_float


class Lib:
    def __init__(self, path):
        from geomconsistentfr_amd import _lib
        self.L = ctypes.CDLL(path)
        self.L.gcfr_shadow_fwd.restype = _i
        self.L.gcfr_shadow_fwd.argtypes = [_p, _p, _i, _p, _i, _i, _i, _i, _i, _p, _f, _p, _p, _p, _p, ctypes.c_size_t, _p, _p]
        self.L.gcfr_shadow_workspace_bytes.restype = ctypes.c_size_t
        self.L.gcfr_shadow_workspace_bytes.argtypes = [_i, _i, _i]
        self.L.gcfr_options_default.argtypes = [_p]
        self.Options = _lib.Options

    def options(self, **kw):
        o = self.Options()
        self.L.gcfr_options_default(ctypes.byref(o))
        for k, v in kw.items():
            setattr(o, k, v)
        return o

    def march(self, sc, want_argmin=True, pixels=0):
        """scene -> (min_dist, argmin) tensors; the grid schedule (no k-split), workspace path"""
        dev = torch.device("cuda:0")
        depth, mask, pt, tt = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"])]
        B, H, W = depth.shape
        L_ = pt.shape[1]
        N = tt.numel()
        md = torch.empty((B, L_, H, W), dtype=torch.float32, device=dev)
        am = torch.empty((B, L_, H, W), dtype=torch.int32, device=dev) if want_argmin else None
        nb = int(self.L.gcfr_shadow_workspace_bytes(B, H, W))
        ws = torch.empty(nb, dtype=torch.uint8, device=dev)
        opt = self.options(ksplit=0, pixels=pixels)
        rc = self.L.gcfr_shadow_fwd(depth.data_ptr(), mask.data_ptr(), mask.shape[0], pt.data_ptr(), B, L_, H, W, N, tt.data_ptr(), 0.0, None,
                                    md.data_ptr(), am.data_ptr() if am is not None else None, ws.data_ptr(), nb, None, ctypes.byref(opt))
        assert rc == 0, rc
        torch.cuda.synchronize()
        return md, am


def main():
    import margin_scenes as MS
    args = sys.argv[1:]
    mutants, fams, seeds, out = None, None, 40, os.path.join(ROOT, "gpurun_out", "hunt.json")
    while args:
        a = args.pop(0)
        if a == "--mutants":
            mutants = [int(x) for x in args.pop(0).split(",")]
        elif a == "--families":
            fams = args.pop(0).split(",")
        elif a == "--seeds":
            seeds = int(args.pop(0))
        elif a == "--out":
            out = args.pop(0)
    if mutants is None:
        mutants = sorted(int(f[4:-3]) for f in os.listdir(LIB_DIR) if f.startswith("mut_") and f.endswith(".so") and f != "mut_0.so")
    fams = fams or sorted(MS.FAMILIES)
    base = Lib(os.path.join(LIB_DIR, "mut_0.so"))
    libs = {n: Lib(os.path.join(LIB_DIR, "mut_%d.so" % n)) for n in mutants}
    res = {str(n): {} for n in mutants}
    t_start = time.time()
    for fam in fams:
        gen = MS.FAMILIES[fam]
        for seed in range(seeds):
            sc = gen(seed)
            for want in (True, False):
                for pixels in ((0, 1) if (want and sc.get("pixels_mask")) else (0,)):
                    md0, am0 = base.march(sc, want, pixels)
                    for n, lib in libs.items():
                        md, am = lib.march(sc, want, pixels)
                        a, b = md.view(torch.int32), md0.view(torch.int32)
                        diff = (a != b)
                        if am is not None:
                            diff |= (am != am0)
                        cnt = int(diff.sum())
                        if cnt:
                            e = res[str(n)].setdefault(fam, {"scenes": 0, "first": []})
                            e["scenes"] += 1
                            if len(e["first"]) < 6:
                                idx = diff.nonzero()[0].tolist()
                                e["first"].append({"seed": seed, "argmin": want, "pixels": pixels, "pixels_differ": cnt, "at": idx,
                                                   "md": [float(md[tuple(idx)]), float(md0[tuple(idx)])]})
        print("family %s done (%.0f s): %s" % (fam, time.time() - t_start,
                                               {n: res[str(n)].get(fam, {}).get("scenes", 0) for n in mutants}), flush=True)
        os.makedirs(os.path.dirname(out), exist_ok=True)
        with open(out, "w") as f:
            json.dump(res, f, indent=1)
    alive = [n for n in mutants if not res[str(n)]]
    print("not killed by any family:", alive)


if __name__ == "__main__":
    main()
