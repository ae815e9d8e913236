from .ico_sphere import ico_sphere
