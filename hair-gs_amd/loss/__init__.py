from .losses import *  # noqa: F401,F403
