import sys, os, numpy as np, torch, time
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import scenes
from dss_amd import ops, _lib
DEV='cuda:0'
pts, nrm = scenes.load_cloud("yoga6"); pts = scenes.normalize_unit_sphere(pts); pts, nrm = scenes.upsample_jitter(pts, nrm, 10, seed=0)
h = scenes.global_h(pts); S,K,thr=512,5,0.05
t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(DEV)
for dist in (2.0, 4.0, 6.0, 11.0):
    M, V, _ = scenes.camera_matrices(dist, 20.0, 30.0)
    P = pts.shape[0]
    args = (t(pts), t(nrm), torch.full((1,), h, device=DEV), t(M), t(V), torch.full((1,), 0.1, device=DEV), torch.full((1,), 100.0, device=DEV),
            torch.zeros(1, dtype=torch.int64, device=DEV), torch.full((1,), P, dtype=torch.int64, device=DEV), torch.ones((P,3), device=DEV), S, K, 1.0, thr, 1.0, False, True)
    for state in (1, 0):
        f = ops.render_forward(*args, workspace_state=state)
        torch.cuda.synchronize()
        t0=time.perf_counter()
        for _ in range(10): ops.render_forward(*args, workspace_state=state)
        torch.cuda.synchronize(); dt=(time.perf_counter()-t0)/10
        tag = ("render_forward" if state == 1 else "render_forward_binned", 1, P, S)
        buf = [b for k,b in _lib._clean_cache.items() if k[2]==tag][0]
        tiles=(S//8)**2; up=lambda x:(x+255)//256*256
        cb=up(tiles*8*4); fb=up(tiles*4)
        w = buf[:3*cb+fb+256].view(torch.int32)
        ctrl = w[(3*cb+fb)//4+32:(3*cb+fb)//4+35].tolist()
        counts = w[:tiles*8]
        print("dist", dist, "state", state, "ms %.3f" % (dt*1e3), "ctrl", ctrl, "max count", int(counts.max()), "nonzero cursor", int((w[cb//4:cb//4+tiles*8]!=0).sum()), "occ", float(f["occupancy"].mean()))
