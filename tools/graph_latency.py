import sys, time, torch, numpy as np
sys.path.insert(0, '/root/repo')
import bench
from geomconsistentfr_amd import RenderParams
from geomconsistentfr_amd import block as R
dev = torch.device('cuda:0')
for B in (1, 2, 8):
    depth, mask, albedo, normals, light, amb = bench.synth_faces(B, 0)
    t = lambda a: torch.from_numpy(a).to(dev)
    args = (t(depth), t(mask), t(light).reshape(B,1,3), t(amb).reshape(B,1), t(normals), t(albedo))
    prm = RenderParams()
    ref = R.render_fwd(*args, prm, want_argmin=False)
    g = R.GraphedRenderFwd(*args, prm)
    out = g(*args)
    torch.cuda.synchronize()
    assert torch.equal(out['rendered_images'], ref['rendered_images']) and torch.equal(out['minimum_distance'], ref['minimum_distance'])
    # new inputs through the graph
    depth2 = t(np.roll(depth, 5, axis=2).copy())
    ref2 = R.render_fwd(depth2, *args[1:], prm, want_argmin=False)
    out2 = g(depth2, *args[1:])
    torch.cuda.synchronize()
    assert torch.equal(out2['rendered_images'], ref2['rendered_images'])
    for name, fn in (('eager', lambda: R.render_fwd(*args, prm, want_argmin=False)), ('graph replay (static inputs)', lambda: g.graph.replay()), ('graph call (copies inputs)', lambda: g(depth2, *args[1:]))):
        for _ in range(20): fn()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(300): fn()
        torch.cuda.synchronize(); dt = (time.perf_counter()-t0)/300
        print('B=%d %-30s %.1f us/step' % (B, name, dt*1e6))
