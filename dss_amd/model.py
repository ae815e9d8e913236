"""The point model of train_mvr.py, mirroring `DSS.models.point_modeling.Model` (DSS/models/point_modeling.py:33-238):
learnable positions / normals / colours of ONE cloud seen by N cameras, shaded by a texture, rendered by the
splatting renderer; `forward` returns the reference's dictionary (`iso_pcl`, `img_pred`, `mask_img_pred`) and leaves
the per-point `visibility` / `inmask` flags of this iteration in `points_filter` for the regularisers.

Everything per-point runs on the HIP path: shading (dss_phong_*), render (dss_render_* / dss_splat_*), and the in-mask
filter (`dss_points_inmask`, one launch instead of the reference's re-projection + grid_sample + any + and).
"""
import torch
import torch.nn.functional as F
from torch import nn

from . import ops
from .cloud import PointClouds3D, PointCloudsFilters


class Model(nn.Module):
    def __init__(self, points, normals, colors, renderer, texture=None, learn_points=True, learn_normals=True,
                 learn_colors=True, device="cpu", **kwargs):
        """points, normals, colors: (1, P, 3); renderer: `SurfaceSplattingRenderer`; texture: `LightingTexture` or
        None (colours are rendered as they are)."""
        super().__init__()
        self.points = nn.Parameter(points.to(device=device)).requires_grad_(learn_points)
        self.normals = nn.Parameter(normals.to(device=device)).requires_grad_(learn_normals)
        self.colors = nn.Parameter(colors.to(device=device)).requires_grad_(learn_colors)
        self.n_points_per_cloud = self.points.shape[1]
        self.renderer = renderer.to(device=device)
        self.texture = texture.to(device=device) if texture is not None else None
        self.register_buffer("points_activation", torch.full(self.points.shape[:2], True, device=device, dtype=torch.bool))
        self.points_filter = PointCloudsFilters(device=device, activation=self.points_activation)
        self.cameras = None  # set in the forward pass

    def decode_color(self, pointclouds, **kwargs):
        return pointclouds if self.texture is None else self.texture(pointclouds, **kwargs)

    def _get_normals(self):
        return F.normalize(self.normals, dim=-1)

    def get_point_clouds(self, points=None, with_colors=False, filter_inactive=True, **kwargs):
        """point_modeling.py:91-115: the cloud of the current parameters (unit normals), optionally without the
        inactive points and with the shaded colours."""
        pointclouds = PointClouds3D(points=self.points if points is None else points, normals=self._get_normals(),
                                    features=self.colors)
        self.points_filter.set_filter(activation=self.points_activation)
        if filter_inactive:
            pointclouds = self.points_filter.filter_with(pointclouds, ("activation",))
        if with_colors:
            pointclouds = self.decode_color(pointclouds, **kwargs)
        return pointclouds

    def _collapse_filters(self):
        """point_modeling.py:172-176: per-camera rows back to one row for the single cloud."""
        flt = self.points_filter
        flt.visibility = flt.visibility.any(dim=0, keepdim=True)
        flt.activation = flt.activation[:1]
        flt.inmask = flt.inmask[:1]

    def forward(self, mask_img=None, **kwargs):
        """-> {'iso_pcl': the (active) cloud for the regularisers, 'img_pred' (N,H,W,3), 'mask_img_pred' (N,H,W,1),
        'rgba_pred' (N,H,W,4)}."""
        self.cameras = kwargs.get("cameras", self.cameras)
        assert self.cameras is not None, "cameras wasn't set."
        batch_size = self.cameras.R.shape[0]
        if batch_size != self.points.shape[0]:
            assert batch_size == 1 or self.points.shape[0] == 1, "Cameras batchsize and points batchsize are incompatible."
        # inactive points are dropped by the renderer (through the filter), not here
        colored = self.get_point_clouds(with_colors=True, filter_inactive=False, **kwargs)
        rgba = self.renderer(colored, point_clouds_filter=self.points_filter, cameras=self.cameras)
        self._collapse_filters()
        rgb, mask = rgba[..., :3], rgba[..., -1:]

        point_clouds = self.get_point_clouds(with_colors=False)
        with torch.no_grad():
            flt = self.points_filter
            if not flt.all_on(flt.activation):       # (1, P) -> (1, P_active); nothing to do when every point is active
                flt.visibility = flt.visibility[flt.activation].unsqueeze(0)
            if mask_img is not None:
                M = self.cameras.get_full_projection_transform().get_matrix().to(self.points.device, torch.float32)
                inmask = ops.points_inmask(point_clouds.points_packed().detach(), M.contiguous(), mask_img,
                                           visible=flt.visibility[0])
                flt.set_filter(inmask=inmask.unsqueeze(0))
        # `rgba_pred` (not in the reference dictionary) is the undivided render: calc_dr_loss takes it as it is, which
        # saves re-concatenating img_pred and mask_img_pred (a 32 MB copy at 8 x 512^2)
        return {"iso_pcl": point_clouds, "img_pred": rgb, "mask_img_pred": mask, "rgba_pred": rgba}

    def render(self, p_world=None, cameras=None, lights=None) -> torch.Tensor:
        """Render the cloud to RGBA (N, H, W, 4) images (point_modeling.py:212-232)."""
        cameras = cameras or self.cameras
        pointclouds = self.get_point_clouds(p_world, with_colors=False, filter_inactive=False)
        colored = self.decode_color(pointclouds, cameras=cameras, lights=lights)
        rgba = self.renderer(colored, point_clouds_filter=self.points_filter, cameras=cameras)
        self._collapse_filters()
        return rgba

    def prune_points(self, mask_gt, loss_func, **kwargs):
        """point_modeling.py:117-138: points whose silhouette-loss gradient is zero in the given views are dead."""
        mask = self.forward(**kwargs)["mask_img_pred"]
        mask_loss = loss_func(mask.squeeze().float(), mask_gt.squeeze().float())
        grad = torch.autograd.grad([mask_loss], [self.points])[0]
        return ~torch.all(grad == 0.0, dim=-1)
