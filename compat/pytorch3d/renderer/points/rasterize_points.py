"""pytorch3d.renderer.points.rasterize_points: only the constant the reference imports (DSS/core/rasterizer.py:22).
pytorch3d's own disc rasterizer is not part of the reference's EWA path and is not provided."""
kMaxPointsPerBin = 22


def rasterize_points(*args, **kwargs):
    raise NotImplementedError("pytorch3d's disc rasterizer is not part of this compatibility layer; the EWA rasterizer is "
                              "dss_amd.rasterizer.SurfaceSplatting")
