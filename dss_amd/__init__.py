"""dss_amd -- MI355X (gfx950) native differentiable EWA surface-splatting rasterizer.

Drop-in for the hot path of yifita/DSS: ``dss_amd.rasterizer.SurfaceSplatting`` /
``PointsRasterizationSettings`` and ``dss_amd.renderer.SurfaceSplattingRenderer`` mirror
``DSS.core.rasterizer`` / ``DSS.core.renderer``; ``dss_amd.ops`` mirrors the native module
``DSS._C``.  All compute runs in hand-written HIP kernels behind the C ABI of
``include/dss_hip.h`` (``dss_amd/csrc/libdss_hip.so``).
"""
__version__ = "0.1.0"
