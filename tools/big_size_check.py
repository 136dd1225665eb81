import sys, time, numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/oracle')
import c_oracle
from geomconsistentfr_amd import RenderParams, light_prep, shadow_min_distance
dev=torch.device('cuda:0')
for (H,W,N,B,L) in [(1024,1024,320,1,2),(1000,1016,160,2,1),(768,300,160,3,2),(2048,512,160,1,1),(4096,4096,8,1,1)]:   # (horizon tables: W % 4 == 0 and H, W <= 1024)
    rng=np.random.default_rng(H+W)
    r,c=np.mgrid[0:H,0:W]
    depth=(0.2*H*np.exp(-(((c-0.5*W)/(0.25*W))**2+((r-0.5*H)/(0.3*H))**2))+rng.random((H,W))).astype(np.float32)[None].repeat(B,0)
    depth[:, :H//8, :] += np.float32(0.3*H)        # a ridge OUTSIDE the mask, higher than the face: the horizon tables must not see it, the samples near the mask's edge do
    mask=((((c-0.5*W)/(0.4*W))**2+((r-0.5*H)/(0.45*H))**2)<1).astype(np.uint8)[None].repeat(B,0)
    lights=rng.standard_normal((B,L,3)).astype(np.float32)
    prm=RenderParams(n_samples=N, dt=0.8/N)
    _,pt=light_prep(torch.from_numpy(lights).to(dev), prm)
    t0=time.time(); md,am=shadow_min_distance(torch.from_numpy(depth).to(dev), torch.from_numpy(mask).to(dev), pt, prm); torch.cuda.synchronize(); tg=time.time()-t0
    _,pto=c_oracle.light_prep(lights.reshape(-1,3), clamp_z_min=0.0)
    t0=time.time(); mdo,amo=c_oracle.shadow_min_distance(depth, mask, pto.reshape(B,L,3), c_oracle.sample_table(0.025,0.8/N,N)); tc=time.time()-t0
    md=md.cpu().numpy(); am=am.cpu().numpy()
    print((H,W,N,B,L), 'bit-equal', np.array_equal(md, mdo), 'argmin diffs', int((am!=np.where(mdo<1e5, amo, -1)).sum()), 'gpu %.3fs cpu %.1fs'%(tg,tc))
