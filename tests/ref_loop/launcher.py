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
The script's working directory is the reference checkout (train_mvr.py:30 loads `configs/default.yaml` relatively); all
outputs go to the out_dir named in the config."""
import argparse
import importlib.abc
import importlib.machinery
import json
import os
import runpy
import sys

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
            self._f.write(json.dumps({"tag": tag, "value": float(value), "step": global_step}) + "\n")
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
    cloud = PointClouds3D(torch.from_numpy(pts)[None].to(dev), torch.from_numpy(nrm)[None].to(dev),
                          torch.ones(1, len(pts), 3, device=dev))
    texture, lights = LightingTexture(), DirectionalLights(device=dev)
    out = args.make_dataset
    os.makedirs(os.path.join(out, "image"), exist_ok=True)
    os.makedirs(os.path.join(out, "mask"), exist_ok=True)
    mats = []
    for i in range(args.views):
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--make-dataset", default=None, help="write a synthetic MVR dataset here instead of training")
    ap.add_argument("--views", type=int, default=16)
    ap.add_argument("--target-points", type=int, default=0)
    ap.add_argument("--reference", default="/root/reference")
    ap.add_argument("--config", required=True)
    ap.add_argument("--exit-after", type=int, default=20)
    ap.add_argument("--no-cuda", action="store_true")
    ap.add_argument("--scalars", required=True, help="JSON-lines file the SummaryWriter stand-in appends to")
    args = ap.parse_args()
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
    if args.make_dataset:
        return make_dataset(args)
    sys.argv = ["train_mvr.py", "--config", args.config, "--exit-after", str(args.exit_after)] + \
        (["--no-cuda"] if args.no_cuda else [])
    runpy.run_path(os.path.join(args.reference, "train_mvr.py"), run_name="__main__")


if __name__ == "__main__":
    main()
