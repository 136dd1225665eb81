import numpy as np, torch, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from geomconsistentfr_amd import RenderParams, _lib
from geomconsistentfr_amd import block as R
from geomconsistentfr_amd.normals import depth_to_normals
rng = np.random.default_rng(11)
B, H, W = 3, 96, 128
dev = torch.device("cuda:0")
depth = torch.from_numpy((30 * rng.random((B, H, W))).astype(np.float32)).to(dev)
mask = torch.from_numpy((rng.random((B, H, W)) > 0.3).astype(np.uint8)).to(dev)
albedo = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).to(dev)
light = torch.from_numpy(rng.standard_normal((B, 2, 3)).astype(np.float32)).to(dev)
amb = torch.from_numpy(rng.random((B, 2), dtype=np.float32)).to(dev)
K = torch.zeros(1, 3, 3, dtype=torch.float64); K[:, 0, 0] = K[:, 1, 1] = 1570.0; K[:, 2, 2] = 1.0; K[:, 0, 2] = W / 2.0; K[:, 1, 2] = H / 2.0
K = K.to(dev)
prm = RenderParams(n_samples=64, dt=0.0125)
cam = (1570.0, 1570.0, W / 2.0, H / 2.0, 1610.0)
n = depth_to_normals(depth[:, None], K, z_offset=1610.0)
a = R.render_fwd(depth, mask, light, amb, n, albedo, prm, want_argmin=True)
def cmp(tag, o, L0=True):
    for k in ("minimum_distance", "shadow_mask_weights", "full_shading", "final_shading", "rendered_images", "surface_normals", "unit_light_direction"):
        if k not in o or o[k] is None: continue
        ref = n if k == "surface_normals" else a[k]
        x = o[k]
        if k != "surface_normals" and x.shape != ref.shape:
            ref = ref[:, :1]
        d = (x != ref)
        if d.any():
            idx = d.nonzero()
            print(tag, k, "differs at", int(d.sum()), "first", idx[0].tolist(), "last", idx[-1].tolist(), float(x[tuple(idx[0])]), float(ref[tuple(idx[0])]))
        else:
            print(tag, k, "equal")
for want in (True, False):
    for Lsel in (2, 1):
        li, am = (light, amb) if Lsel == 2 else (light[:, :1].contiguous(), amb[:, :1].contiguous())
        o = R.render_fwd(depth, mask, li, am, None, albedo, prm, want_argmin=want, camera=cam)
        cmp("fused argmin=%s L=%d" % (want, Lsel), o)
        o = R.render_fwd(depth, mask, li, am, n, albedo, prm, want_argmin=want)
        cmp("normals-in argmin=%s L=%d" % (want, Lsel), o)
with torch.no_grad():
    r = R.render_from_depth(depth[:, None], albedo, light[:, 0], amb[:, 0], K, 1610.0, mask, prm)
d = r["rendered_images"] != a["rendered_images"][:, 0]
print("render_from_depth rendered differs", int(d.sum()), d.nonzero()[:3].tolist())
print("ambient_values", r["ambient_values"].flatten().tolist(), amb[:, 0].tolist())
print("full", float((r["full_shading"] != a["full_shading"][:, 0]).sum()), "w", float((r["shadow_mask_weights"] != a["shadow_mask_weights"][:, 0]).sum()))
