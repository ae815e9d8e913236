from .transform3d import Rotate, RotateAxisAngle, Scale, Transform3d, Translate

__all__ = ["Rotate", "RotateAxisAngle", "Scale", "Transform3d", "Translate"]
