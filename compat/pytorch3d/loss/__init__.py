from .chamfer import chamfer_distance
