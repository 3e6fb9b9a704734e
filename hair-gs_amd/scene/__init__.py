"""Scene-side containers of the hot path.  (The reference's `Scene` loader -- COLMAP IO, PLY resume,
scene/__init__.py:30-134 -- is dataset plumbing outside the accelerated path; synthetic scenes are built by
`synthetic.*`.)"""
from .cameras import Camera, MiniCam  # noqa: F401
from .gaussian_model import GaussianModel  # noqa: F401
from .hair_gaussian_model import HairGaussianModel, StrandsInfo  # noqa: F401
