"""Generates the committed fixtures under tests/golden/ (run HERE, in the build container, where
/root/reference and oracle/_ref exist; the GPU box only reads the .npz files).

    python tests/golden/make_golden.py

* clouds.npz            point clouds converted from the reference's PLY fixtures
                        (example_data/pointclouds/{teapot_normal_dense,bunny-8000,point-one,yoga6_out}.ply)
* ref_teapot256.npz     BASELINE config 1: teapot, 1 camera, 256x256, K=5 -- inputs + the outputs of
                        the UNMODIFIED reference CPU rasterizer (oracle/_ref: splat_points bin_size=0,
                        _splat_points_occ_backward, _backward_zbuf)
* ref_random48.npz      1500 random anisotropic splats, 48x48, N=1 (7 points behind the camera)
* ref_random64x2.npz    2 clouds x 800 splats, 64x64, K=3
* ref_ties32.npz        depth-tie stress case (quantised z), 32x32, K=4
"""
import os
import sys


sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.dirname(HERE))

import oracle  # noqa: E402
import scenes  # noqa: E402

REF_PLY = "/root/reference/example_data/pointclouds"


def read_ply(path):
    with open(path, "rb") as f:
        header = []
        while True:
            line = f.readline().decode("ascii").strip()
            header.append(line)
            if line == "end_header":
                break
        nv = [int(l.split()[-1]) for l in header if l.startswith("element vertex")][0]
        props = []
        in_vertex = False
        for l in header:
            if l.startswith("element"):
                in_vertex = l.startswith("element vertex")
            elif l.startswith("property") and in_vertex:
                props.append(l.split()[1:])
        if any("binary_little_endian" in l for l in header):
            dt = np.dtype([(p[1], {"float": "<f4", "uchar": "u1", "int": "<i4"}[p[0]]) for p in props])
            data = np.frombuffer(f.read(nv * dt.itemsize), dtype=dt, count=nv)
            cols = {n: data[n].astype(np.float32) for n in dt.names}
        else:
            arr = np.loadtxt(f, max_rows=nv, ndmin=2)
            cols = {p[1]: arr[:, i].astype(np.float32) for i, p in enumerate(props)}
    pts = np.stack([cols["x"], cols["y"], cols["z"]], 1)
    nrm = np.stack([cols["nx"], cols["ny"], cols["nz"]], 1)
    return pts, nrm


def run_ref(sc, K, thr, radii_s, seed):
    R = oracle.ref()
    assert R is not None, "build oracle/_ref first: make -C oracle ref"
    t = lambda k: torch.from_numpy(np.ascontiguousarray(sc[k]))
    S, N, P = sc["S"], sc["first_idx"].shape[0], sc["points"].shape[0]
    idx, zbuf, qv, occ = R.splat_points(t("points"), t("ellipse"), t("cutoff"), t("radii"), t("first_idx"),
                                        t("num_pts"), thr, S, K, 0, 0)
    rng = np.random.default_rng(seed)
    grad_occ = rng.standard_normal((N, S, S)).astype(np.float32)
    grad_occ[rng.random((N, S, S)) < 0.3] = 0
    grad_zbuf = rng.standard_normal((N, S, S, K)).astype(np.float32)
    gslow = R._splat_points_occ_backward(t("points"), t("radii"), torch.from_numpy(grad_occ), t("first_idx"),
                                         t("num_pts"), radii_s, thr)
    gz = torch.zeros(P, 1)
    R._backward_zbuf(idx, torch.from_numpy(grad_zbuf), gz)
    out = {k: sc[k] for k in ("points", "ellipse", "cutoff", "radii", "scaler", "colors", "first_idx", "num_pts")}
    out.update(S=np.int32(S), K=np.int32(K), thr=np.float32(thr), radii_s=np.float32(radii_s),
               ref_idx=idx.numpy(), ref_zbuf=zbuf.numpy(), ref_qvalue=qv.numpy(), ref_occ=occ.numpy(),
               grad_occ=grad_occ, grad_zbuf=grad_zbuf, ref_grad_occ_slow=gslow.numpy(), ref_grad_z=gz.numpy()[:, 0])
    return out


def main():
    clouds = {}
    for key, fn in (("teapot", "teapot_normal_dense.ply"), ("bunny", "bunny-8000.ply"), ("one", "point-one.ply"),
                    ("yoga6", "yoga6_out.ply")):
        p, n = read_ply(os.path.join(REF_PLY, fn))
        clouds[key + "_points"], clouds[key + "_normals"] = p, n
        print(key, p.shape)
    np.savez_compressed(os.path.join(HERE, "clouds.npz"), **clouds)

    # config 1: teapot, 1 camera (fov 60, znear 0.1, dist 2, elev 30, azim 45), 256^2, K=5
    pts = scenes.normalize_unit_sphere(clouds["teapot_points"])
    M, V, _ = scenes.camera_matrices(2.0, 30.0, 45.0)
    sc = scenes.setup_scene(pts, clouds["teapot_normals"], M, V, 256)
    np.savez_compressed(os.path.join(HERE, "ref_teapot256.npz"), **run_ref(sc, 5, 0.05, 5.0, 1))

    np.savez_compressed(os.path.join(HERE, "ref_random48.npz"),
                        **run_ref(scenes.random_splats(1500, 48, 1, seed=0, negz=7), 5, 0.05, 5.0, 2))
    np.savez_compressed(os.path.join(HERE, "ref_random64x2.npz"),
                        **run_ref(scenes.random_splats(800, 64, 2, seed=1), 3, 0.05, 3.0, 3))
    np.savez_compressed(os.path.join(HERE, "ref_ties32.npz"),
                        **run_ref(scenes.random_splats(600, 32, 1, seed=2, ties=True), 4, 0.3, 2.0, 4))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)) // 1024, "KiB")


if __name__ == "__main__":
    main()
