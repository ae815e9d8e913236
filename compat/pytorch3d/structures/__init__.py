from .meshes import Meshes, join_meshes_as_batch
from .pointclouds import Pointclouds
from .utils import list_to_packed, list_to_padded, packed_to_list, padded_to_list, padded_to_packed

__all__ = ["Meshes", "Pointclouds", "join_meshes_as_batch", "list_to_packed", "list_to_padded", "packed_to_list",
           "padded_to_list", "padded_to_packed"]
