from .losses import *  # noqa: F401,F403
from .metrics import *  # noqa: F401,F403
