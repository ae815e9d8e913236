import sys, numpy as np
sys.path.insert(0,'/root/repo/tests'); sys.path.insert(0,'/root/repo')
import oracle, scenes
pts, nrm = scenes.load_cloud("bunny"); pts = scenes.normalize_unit_sphere(pts)
S=128
M,V,_ = scenes.camera_matrices(2.0,30.0,45.0)
sc = scenes.setup_scene(pts, nrm, M, V, S, h=scenes.global_h(pts))
idx,zb,qv,occ = oracle.splat_forward(sc["points"], sc["ellipse"], sc["cutoff"], sc["radii"], sc["first_idx"], sc["num_pts"], S, 5, 0.05)
rng=np.random.default_rng(0)
go = (rng.standard_normal((1,S,S))/ (S*S)).astype(np.float32)
g0,_,rs = oracle.splat_backward(sc["points"], sc["radii"], idx, go, None, sc["first_idx"], sc["num_pts"], 5.0, 0.05)
for ulps in (1,2):
    p2 = sc["points"].copy()
    sign = rng.integers(-ulps, ulps+1, p2.shape).astype(np.int32)
    p2 = (p2.view(np.int32) + sign).view(np.float32)
    g1,_,_ = oracle.splat_backward(p2, sc["radii"], idx, go, None, sc["first_idx"], sc["num_pts"], 5.0, 0.05)
    d = g1-g0
    print("perturbation of +-%d ulp on pts_screen: rel-L2 of the clipped surrogate gradient %.2e, max abs diff %.2e of max %.2e, points that changed by > 1e-6 rel: %d of %d" % (ulps, np.linalg.norm(d)/np.linalg.norm(g0), np.abs(d).max(), np.abs(g0).max(), int((np.abs(d).max(1) > 1e-6*np.abs(g0).max()).sum()), g0.shape[0]))
