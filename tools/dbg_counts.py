"""work counters of the march on margin scenes: product counting build vs mutant counting builds (diagnosis)"""
import ctypes, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import margin_scenes as MS
from mutant_hunt import Lib
from geomconsistentfr_amd import _lib
names = _lib.COUNTER_NAMES
def run(lib, sc, want):
    dev = torch.device("cuda:0")
    depth, mask, pt, tt = [torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (sc["depth"], sc["mask"], sc["light_pt"], sc["t_table"])]
    B, H, W = depth.shape
    md = torch.empty((B, 1, H, W), dtype=torch.float32, device=dev)
    am = torch.empty((B, 1, H, W), dtype=torch.int32, device=dev) if want else None
    nb = int(lib.L.gcfr_shadow_workspace_bytes(B, H, W)); ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    tiles = B * ((W + 15) // 16) * ((H + 3) // 4)
    cnt = torch.zeros(20 + 4 * tiles, dtype=torch.int64, device=dev)
    opt = lib.options(ksplit=0, counters=cnt.data_ptr())
    rc = lib.L.gcfr_shadow_fwd(depth.data_ptr(), mask.data_ptr(), B, pt.data_ptr(), B, 1, H, W, tt.numel(), tt.data_ptr(), 0.0, None, md.data_ptr(),
                               am.data_ptr() if want else None, ws.data_ptr(), nb, None, ctypes.byref(opt))
    assert rc == 0
    torch.cuda.synchronize()
    c = cnt[:20].cpu().numpy()
    return {n: int(c[i]) for i, n in enumerate(names)}, md
libs = {n: Lib(os.path.join(ROOT, "geomconsistentfr_amd", "lib", n + ".so")) for n in sys.argv[1].split(",")}
for fam, seed in [x.split(":") for x in sys.argv[2].split(",")]:
    sc = MS.FAMILIES[fam](int(seed))
    base = None
    for n, lib in libs.items():
        c, md = run(lib, sc, False)
        keep = {k: c[k] for k in ("tiles", "groups_visited", "bound_tests", "bodies", "early_exits", "bounds_given_up", "trail_enter", "trail_skips", "trail_leave", "rough_samples")}
        d = "" if base is None else " differs %d" % int((md.view(torch.int32) != base.view(torch.int32)).sum())
        base = md if base is None else base
        print(fam, seed, n, keep, d)
