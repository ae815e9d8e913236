#!/usr/bin/env python3
"""Runs the reference's OWN `train_mvr.py` (unmodified, from /root/reference) on top of the drop-in classes.

    python tests/ref_loop/launcher.py --reference /root/reference --config cfg.yml --exit-after 20 [--no-cuda]

What this file adds around the untouched script (none of it is shipped; it is what a maintainer's environment provides):
  * sys.path: compat/ (pytorch3d + easydict stand-ins), the repo (dss_amd), the reference checkout;
  * import stand-ins for packages the reference imports at module level but this image lacks and the loop does not
    need: git (train_mvr.py:4,60-62 logs the commit hash), imageio (dataset.py:5 -> PIL), trimesh / skimage / plyfile /
    pymeshlab / frnn / torch_batch_svd / prefix_sum (mesh export, FRNN grid of the reference rasterizer -- replaced),
    tensorboard (SummaryWriter -> JSON lines in the log directory, which the test reads back);
  * two aliases removed from the libraries since the reference was written: `np.bool` (dataset.py:99), `torch._six`;
  * on a machine WITHOUT a GPU only: `dss_amd.ops` answered by the oracle (tests/ref_loop/oracle_ops.py), because the
    product refuses to run without the HIP library.  With a GPU the real kernels run.
Developer aid: DSS_REF_LOOP_CPROFILE=<file> runs the script under cProfile and writes the top 60 entries (sorted by
DSS_REF_LOOP_CPROFILE_SORT, default `cumulative`) plus the callers of .cpu() / .item() / .tolist() -- the requests that drain
the GPU queue -- to that file.
The script's working directory is the reference checkout (train_mvr.py:30 loads `configs/default.yaml` relatively); all
outputs go to the out_dir named in the config."""
import argparse
import importlib.abc
import importlib.machinery
import json
import os
import runpy
import sys
import time

sys.dont_write_bytecode = True  # the reference checkout is read-only: importing it must not leave __pycache__ there
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


class _Lazy(types.ModuleType):
    """module whose every attribute is a callable placeholder that fails when actually used"""

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        full = "%s.%s" % (self.__name__, name)

        def missing(*a, **k):
            raise RuntimeError("%s is not available in this environment (import stand-in)" % full)
        missing.__name__ = name
        return missing


class _StubFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    ROOTS = ("trimesh", "skimage", "plyfile", "pymeshlab", "frnn", "torch_batch_svd", "prefix_sum", "dominate",
             "torchvision", "cv2", "tensorboard")

    def find_spec(self, fullname, path=None, target=None):
        if fullname.split(".")[0] in self.ROOTS:
            return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
        return None

    def create_module(self, spec):
        m = _Lazy(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


def _install_stand_ins(log_path):
    sys.meta_path.append(_StubFinder())
    import numpy as np
    import torch
    if not hasattr(np, "bool"):
        np.bool = bool  # removed in numpy 1.24 (dataset.py:99)
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "int"):
        np.int = int
    # `mask[mask] = values` (rasterizer.py:641) indexed a tensor with itself: legal in the torch the reference was written
    # for, refused by current torch ("input tensor and written-to tensor refer to a single memory location")
    _setitem = torch.Tensor.__setitem__

    def _setitem_self_index(self, idx, val):
        return _setitem(self, idx.clone() if idx is self else idx, val)
    torch.Tensor.__setitem__ = _setitem_self_index
    six = types.ModuleType("torch._six")
    six.string_classes = (str, bytes)
    six.int_classes = (int,)
    six.container_abcs = __import__("collections").abc
    six.inf = float("inf")
    sys.modules.setdefault("torch._six", six)

    git = types.ModuleType("git")

    class Repo:  # train_mvr.py:60-62
        def __init__(self, *a, **k):
            self.head = types.SimpleNamespace(object=types.SimpleNamespace(hexsha="0" * 40))
    git.Repo = Repo
    sys.modules["git"] = git

    imageio = types.ModuleType("imageio")

    def imread(path, pilmode=None, **kwargs):
        from PIL import Image
        im = Image.open(path)
        if pilmode is not None:
            im = im.convert(pilmode)
        return np.asarray(im)

    def imwrite(path, arr, **kwargs):
        from PIL import Image
        a = np.asarray(arr)
        if a.dtype != np.uint8:
            a = (np.clip(a, 0, 1) * 255).astype(np.uint8)
        Image.fromarray(a).save(path)
    imageio.imread, imageio.imwrite, imageio.imsave = imread, imwrite, imwrite
    sys.modules["imageio"] = imageio

    tb = types.ModuleType("torch.utils.tensorboard")

    class SummaryWriter:
        """records scalars as JSON lines (the test reads the loss curve back)"""

        def __init__(self, log_dir=None, *a, **k):
            self._f = open(log_path, "a")

        def add_scalar(self, tag, value, global_step=None, *a, **k):
            self._f.write(json.dumps({"tag": tag, "value": float(value), "step": global_step, "t": time.time()}) + "\n")
            self._f.flush()

        def add_scalars(self, main_tag, d, global_step=None, *a, **k):
            for kk, v in d.items():
                self.add_scalar("%s/%s" % (main_tag, kk), v, global_step)

        def __getattr__(self, name):
            if name.startswith("add_") or name in ("flush", "close"):
                return lambda *a, **k: None
            raise AttributeError(name)
    tb.SummaryWriter = SummaryWriter
    sys.modules["torch.utils.tensorboard"] = tb
    import torch.utils
    torch.utils.tensorboard = tb


def make_dataset(args):
    """Synthetic multi-view dataset in the reference's MVR layout (DSS/utils/dataset.py:16-210), rendered by the
    reference's own LightingTexture + lights through the renderer that `config.create_renderer` builds from the YAML."""
    import numpy as np
    import torch
    from PIL import Image
    import config  # the reference's
    from DSS.core.cloud import PointClouds3D
    from DSS.core.lighting import DirectionalLights
    from DSS.core.texture import LightingTexture
    from pytorch3d.renderer import FoVPerspectiveCameras, look_at_view_transform
    cfg = config.load_config(args.config, "configs/default.yaml")
    dev = torch.device("cpu" if args.no_cuda or not torch.cuda.is_available() else "cuda")
    renderer = config.create_renderer(cfg.renderer).to(dev)
    z = np.load(os.path.join(ROOT, "tests", "golden", "clouds.npz"))
    pts, nrm = z["yoga6_points"].astype(np.float32), z["yoga6_normals"].astype(np.float32)
    pts = pts - 0.5 * (pts.max(0) + pts.min(0))
    pts = pts * (0.45 / np.linalg.norm(pts, axis=1).max())
    if args.target_points and args.target_points < len(pts):
        keep = np.random.default_rng(0).permutation(len(pts))[: args.target_points]
        pts, nrm = pts[keep], nrm[keep]
    if args.jitter > 1:   # BASELINE configs[2]: the 9,979-point scan upsampled x10 on its tangent planes = 99,790 points
        import scenes
        pts, nrm = scenes.upsample_jitter(pts, nrm, args.jitter, seed=0)
    cloud = PointClouds3D(torch.from_numpy(pts)[None].to(dev), torch.from_numpy(nrm)[None].to(dev),
                          torch.ones(1, len(pts), 3, device=dev))
    texture, lights = LightingTexture(), DirectionalLights(device=dev)
    out = args.make_dataset
    os.makedirs(os.path.join(out, "image"), exist_ok=True)
    os.makedirs(os.path.join(out, "mask"), exist_ok=True)
    mats = []
    if args.camera_sampler:   # the reference's own sampling rule (DSS/core/camera.py:41-51), unmodified
        from DSS.core.camera import CameraSampler
        torch.manual_seed(0)
        sampler = CameraSampler(args.views, 1, distance_range=torch.tensor([[1.55, 2.0]]), sort_distance=False)
    for i in range(args.views):
        if args.camera_sampler:
            R, T = sampler.R[i:i + 1], sampler.T[i:i + 1]
        else:
            R, T = look_at_view_transform(1.6, 25.0 * np.sin(1.7 * i), 360.0 * i / args.views)
        cams = FoVPerspectiveCameras(R=R, T=T, device=dev)
        with torch.no_grad():
            rgba = renderer(texture(cloud, cameras=cams, lights=lights), cameras=cams)[0].clamp(0, 1).cpu().numpy()
        Image.fromarray((rgba[..., :3] * 255).astype(np.uint8)).save(os.path.join(out, "image", "%03d.png" % i))
        Image.fromarray(((rgba[..., 3] > 0) * 255).astype(np.uint8)).save(os.path.join(out, "mask", "%03d.png" % i))
        mats.append(cams.get_world_to_view_transform().get_matrix()[0].cpu().numpy())
    np.savez(os.path.join(out, "data_dict.npz"), camera_mat=np.stack(mats).astype(np.float32), points=pts, normals=nrm,
             colors=np.ones_like(pts), lights_type="DSS.core.lighting.DirectionalLights",
             cameras_type="pytorch3d.renderer.FoVPerspectiveCameras", cameras_params={})
    print("dataset:", out, "views", args.views, "coverage", float((rgba[..., 3] > 0).mean()))


def make_checkpoint(args):
    """A `model.pt` in the format of the reference's CheckpointIO (DSS/misc/checkpoints.py:28-41, loaded with
    strict=False by train_mvr.py:98-103 through `resume_from: model.pt`): the dataset's own cloud with Gaussian noise on the
    positions and normals -- the unmodified script then RESUMES from it (refinement of a noisy scan instead of the
    sphere of config.py:177-183)."""
    import numpy as np
    import torch
    d = np.load(os.path.join(args.data_dir, "data_dict.npz"), allow_pickle=True)
    rng = np.random.default_rng(1)
    pts = d["points"].astype(np.float32)
    nrm = d["normals"].astype(np.float32)
    pts = pts + rng.normal(0, args.noise, pts.shape).astype(np.float32)
    nrm = nrm + rng.normal(0, 0.5, nrm.shape).astype(np.float32)
    nrm /= np.maximum(np.linalg.norm(nrm, axis=1, keepdims=True), 1e-12)
    os.makedirs(os.path.dirname(args.make_checkpoint), exist_ok=True)
    torch.save({"model": {"points": torch.from_numpy(pts)[None], "normals": torch.from_numpy(nrm)[None]},
                "epoch_it": -1, "it": -1}, args.make_checkpoint)
    print("checkpoint:", args.make_checkpoint, pts.shape)


def _install_frnn_stand_ins():
    """lxxue/FRNN and lxxue/prefix_sum (CUDA extensions the reference's Python calls, absent here) by their published
    behaviour: 2-D grid insertion (cell = floor((p - min) * delta), linear id x * res_y + y, slot = arrival order),
    exclusive scan, counting sort; `frnn_grid_points` = K nearest neighbours within radius r (-1 beyond)."""
    import torch
    import frnn
    import prefix_sum

    def insert_points(pts2d, lengths, grid_params, cnt, cell, slot, G):
        for n in range(pts2d.shape[0]):   # vectorised per cloud (100k visible points per view at configs[2])
            L = int(lengths[n])
            gp = grid_params[n]
            g = torch.floor((pts2d[n, :L] - gp[0:2][None]) * gp[2]).long()
            g = torch.minimum(g.clamp_min(0), (gp[3:5].long() - 1)[None])
            c = g[:, 0] * int(gp[4]) + g[:, 1]
            counts = torch.bincount(c, minlength=G)
            order = torch.sort(c, stable=True)[1]                 # arrival order inside a cell = point order
            first = torch.cumsum(counts, 0) - counts
            s_ = torch.empty(L, dtype=torch.int64, device=c.device)
            s_[order] = torch.arange(L, device=c.device) - first[c[order]]
            cnt[n] = counts.int()
            cell[n, :L] = c.int()
            slot[n, :L] = s_.int()

    def prefix_sum_cuda(counts, total, out):
        t = int(total)
        c = counts[:t].long()
        out[:t] = (torch.cumsum(c, 0) - c).to(out.dtype)

    def counting_sort(pts2d, lengths, cell, slot, off, sorted_pts, sorted_idx):
        for n in range(pts2d.shape[0]):
            L = int(lengths[n])
            dst = (off[n][cell[n, :L].long()] + slot[n, :L]).long()
            sorted_pts[n, dst] = pts2d[n, :L]
            sorted_idx[n, dst] = torch.arange(L, dtype=sorted_idx.dtype, device=sorted_idx.device)

    def frnn_grid_points(p1, p2, lengths1=None, lengths2=None, K=-1, r=-1, grid=None, return_nn=False, return_sorted=True,
                         radius_cell_ratio=2.0):
        from pytorch3d.ops import knn_points
        out = knn_points(p1, p2, lengths1, lengths2, K=K, return_nn=return_nn)
        far = out.dists > float(r) * float(r)
        dists = torch.where(far, torch.full_like(out.dists, -1.0), out.dists)
        idx = torch.where(far, torch.full_like(out.idx, -1), out.idx)
        return dists, idx, out.knn, None
    frnn._C = types.SimpleNamespace(insert_points_cuda=insert_points, counting_sort_cuda=counting_sort)
    frnn.frnn_grid_points = frnn_grid_points
    prefix_sum.prefix_sum_cuda = prefix_sum_cuda


def check_c_seam(args):
    """The reference's OWN rasterizer + renderer classes (DSS/core/rasterizer.py, renderer.py, unmodified) on top of
    `DSS._C = dss_amd.ops` -- the seven same-name mirrors of the compiled extension (ext.cpp:5-18) -- against the
    drop-in classes on the same cloud and cameras.  Forward only: the reference's backward needs FRNN / prefix_sum."""
    import numpy as np
    import torch
    from dss_amd import ops
    import DSS
    DSS._C = ops
    sys.modules["DSS._C"] = ops
    import DSS.core.rasterizer as R           # `from .. import _C` now resolves to the mirror
    from DSS.core.renderer import SurfaceSplattingRenderer as RefRenderer
    from DSS.core.cloud import PointClouds3D
    from pytorch3d.renderer import FoVPerspectiveCameras, NormWeightedCompositor, look_at_view_transform
    import dss_amd.rasterizer as ours_r
    import dss_amd.renderer as ours_rr
    dev = torch.device("cpu" if args.no_cuda or not torch.cuda.is_available() else "cuda")
    z = np.load(os.path.join(ROOT, "tests", "golden", "clouds.npz"))
    pts, nrm = z["bunny_points"].astype(np.float32), z["bunny_normals"].astype(np.float32)
    pts = pts - 0.5 * (pts.max(0) + pts.min(0))
    pts = pts / np.linalg.norm(pts, axis=1).max()
    nrm = nrm / np.linalg.norm(nrm, axis=1, keepdims=True)
    pts, nrm = pts[::2].copy(), nrm[::2].copy()
    g = torch.Generator().manual_seed(0)
    col = torch.rand((1, len(pts), 3), generator=g).to(dev)
    cloud = PointClouds3D(torch.from_numpy(pts)[None].to(dev), torch.from_numpy(nrm)[None].to(dev), col)
    Rm, T = look_at_view_transform((2.0, 2.3), (20.0, -15.0), (30.0, 210.0))
    # cameras the way DSS builds them: one camera object whose R / T are then replaced by the batch's
    # (dataset.py:155-165, trainer.py:264-265) -- znear / zfar stay (1,), which rasterizer.py:190-192 relies on
    cams = FoVPerspectiveCameras(znear=0.1, zfar=100.0, device=dev)
    cams.R, cams.T = Rm.to(dev), T.to(dev)
    cams._N = cams.R.shape[0]
    kw = dict(backface_culling=False, Vrk_invariant=True, Vrk_isotropic=False, cutoff_threshold=1.0,
              depth_merging_threshold=0.05, image_size=96, points_per_pixel=5, radii_backward_scaler=5, clip_pts_grad=0.05)
    ref_ras = R.SurfaceSplatting(cameras=cams, raster_settings=R.PointsRasterizationSettings(**kw), frnn_radius=-1)
    our_ras = ours_r.SurfaceSplatting(cameras=cams, raster_settings=ours_r.PointsRasterizationSettings(**kw))
    with torch.no_grad():
        f_ref, pc_ref = ref_ras(cloud)
        f_our, pc_our = our_ras(cloud)
        img_ref = RefRenderer(ref_ras, NormWeightedCompositor())(cloud)
        img_our = ours_rr.SurfaceSplattingRenderer(our_ras, ours_rr.NormWeightedCompositor())(cloud)
    # ---- backward: the reference's whole EllipticalRasterizer.backward (rasterizer.py:787-977) with
    # _C._splat_points_occ_fast_cuda_backward / _C._backward_zbuf = the mirrors.  Its FRNN grid build needs lxxue/FRNN and
    # lxxue/prefix_sum (absent): stand-ins of their published behaviour (2-D cell = floor((p - min) * delta), linear id
    # x * res_y + y, slot = arrival order; exclusive scan; counting sort) -- the mirror ignores the grid anyway.
    _install_frnn_stand_ins()
    gimg = torch.randn((2, 96, 96, 4), generator=torch.Generator().manual_seed(3)).to(dev)

    def grads(renderer):
        P3 = torch.from_numpy(pts)[None].to(dev).requires_grad_(True)
        C3 = col.clone().requires_grad_(True)
        img = renderer(PointClouds3D(P3, torch.from_numpy(nrm)[None].to(dev), C3))
        (img * gimg).sum().backward()
        return P3.grad[0], C3.grad[0]
    gp_ref, gc_ref = grads(RefRenderer(ref_ras, NormWeightedCompositor()))
    gp_our, gc_our = grads(ours_rr.SurfaceSplattingRenderer(our_ras, ours_rr.NormWeightedCompositor()))
    rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
    same = (f_ref.idx == f_our.idx)
    hit = (f_ref.idx >= 0) & (f_our.idx >= 0) & same
    out = {"idx_equal_fraction": float(same.float().mean()), "fragments": int((f_ref.idx >= 0).sum()),
           "occupancy_equal_fraction": float((f_ref.occupancy == f_our.occupancy).float().mean()),
           "qvalue_max_abs_diff_on_equal": float((f_ref.qvalue - f_our.qvalue)[hit].abs().max()),
           "zbuf_max_abs_diff_on_equal": float((f_ref.zbuf - f_our.zbuf)[hit].abs().max()),
           "scaler_rel_max_diff_on_equal": float(((f_ref.scaler - f_our.scaler)[hit].abs() /
                                                  f_ref.scaler[hit].abs().clamp_min(1e-12)).max()),
           "image_max_abs_diff": float((img_ref - img_our).abs().max()), "image_mean_abs_diff": float((img_ref - img_our).abs().mean()),
           "same_cloud_returned": bool(torch.equal(pc_ref.points_packed(), pc_our.points_packed())),
           "coverage": float((f_ref.occupancy > 0).float().mean()),
           "grad_points_rel_l2": rel(gp_our, gp_ref), "grad_colors_rel_l2": rel(gc_our, gc_ref),
           "grad_points_norm": float(gp_ref.norm()), "grad_points_finite": bool(torch.isfinite(gp_ref).all())}
    print("C_SEAM " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check-c-seam", action="store_true", help="run the reference's own rasterizer classes on DSS._C = dss_amd.ops")
    ap.add_argument("--c-level", nargs="?", const="ops", default=None, choices=["ops", "stub"],
                    help="training / dataset modes: leave the YAML class paths alone and provide DSS._C instead: "
                         "dss_amd.ops (default) or integration/DSS_C.py (`stub`, GPU only)")
    ap.add_argument("--make-dataset", default=None, help="write a synthetic MVR dataset here instead of training")
    ap.add_argument("--make-checkpoint", default=None, help="write a CheckpointIO-format model.pt (noisy copy of the "
                    "dataset's cloud) here instead of training; needs --data-dir")
    ap.add_argument("--data-dir", default=None)
    ap.add_argument("--noise", type=float, default=0.01)
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--target-points", type=int, default=0)
    ap.add_argument("--jitter", type=int, default=1, help="upsample the target cloud by this factor (tangent-plane jitter)")
    ap.add_argument("--camera-sampler", action="store_true", help="draw the views with the reference's CameraSampler")
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--config", default=None)
    ap.add_argument("--exit-after", type=int, default=20)
    ap.add_argument("--no-cuda", action="store_true")
    ap.add_argument("--scalars", default=os.devnull, help="JSON-lines file the SummaryWriter stand-in appends to")
    args = ap.parse_args()
    if "RANK" in os.environ and args.scalars != os.devnull:
        # under torchrun (the row-sharded renderer: every rank runs the same script on the same batches): one scalars file per rank
        args.scalars = "%s.rank%d" % (args.scalars, int(os.environ["RANK"]))
    for p in (os.path.join(ROOT, "compat"), ROOT, os.path.join(ROOT, "tests"), args.reference):
        if p not in sys.path:
            sys.path.insert(0, p)
    _install_stand_ins(args.scalars)
    import torch
    if args.no_cuda or not torch.cuda.is_available():
        from dss_amd import ops
        sys.path.insert(0, HERE)
        import oracle_ops
        oracle_ops.install(ops)
    os.chdir(args.reference)
    if args.c_level:
        if args.c_level == "stub":   # the ctypes file of INTEGRATION.md section 3 (needs a GPU: it calls the library directly)
            import importlib.util
            spec = importlib.util.spec_from_file_location("DSS_C_stub", os.path.join(ROOT, "integration", "DSS_C.py"))
            _ops = importlib.util.module_from_spec(spec)
            spec.loader.exec_module(_ops)
        else:
            from dss_amd import ops as _ops
        import DSS
        DSS._C = _ops
        sys.modules["DSS._C"] = _ops
        _install_frnn_stand_ins()
    if args.check_c_seam:
        return check_c_seam(args)
    if args.make_dataset:
        return make_dataset(args)
    if args.make_checkpoint:
        return make_checkpoint(args)
    sys.argv = ["train_mvr.py", "--config", args.config, "--exit-after", str(args.exit_after)] + \
        (["--no-cuda"] if args.no_cuda else [])
    prof_out = os.environ.get("DSS_REF_LOOP_CPROFILE")   # developer aid: host-side profile of the loop (cumulative, top 60) into this file
    if not prof_out:
        runpy.run_path(os.path.join(args.reference, "train_mvr.py"), run_name="__main__")
        return
    import cProfile
    import io
    import pstats
    pr = cProfile.Profile()
    try:
        pr.runcall(runpy.run_path, os.path.join(args.reference, "train_mvr.py"), run_name="__main__")
    finally:
        buf = io.StringIO()
        st = pstats.Stats(pr, stream=buf).sort_stats(os.environ.get("DSS_REF_LOOP_CPROFILE_SORT", "cumulative"))
        st.print_stats(60)
        st.print_stats(ROOT.replace(os.sep, "/") + "/", 40)   # the same ordering, this repository's files only
        st.print_callers("'(cpu|item|tolist|nonzero)' of")   # who asks the device for a value (each such call drains the queue)
        with open(prof_out, "w") as f:
            f.write(buf.getvalue())


if __name__ == "__main__":
    main()
