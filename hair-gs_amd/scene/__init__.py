"""Scene-side containers of the hot path and, around them, the dataset / on-disk plumbing of SURVEY.md 8f n4
(`Scene`: COLMAP capture + PLY resume, reference scene/__init__.py:30-134); synthetic scenes are built by `synthetic.*`."""
from .cameras import Camera, MiniCam  # noqa: F401
from .gaussian_model import GaussianModel  # noqa: F401
from .hair_gaussian_model import HairGaussianModel, StrandsInfo  # noqa: F401


def __getattr__(name):   # `from scene import Scene` without importing PIL / the dataset readers for every model user
    if name == "Scene":
        from .scene import Scene
        return Scene
    raise AttributeError(name)
