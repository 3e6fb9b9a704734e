from .camera import *  # noqa: F401,F403
from .general import *  # noqa: F401,F403
from .graphics import *  # noqa: F401,F403
from .sh import *  # noqa: F401,F403
from .transform import *  # noqa: F401,F403
