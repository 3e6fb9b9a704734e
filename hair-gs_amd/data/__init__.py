"""Dataset IO around the hot path: COLMAP sparse models and the camera infos built from them."""
from data.colmap import *  # noqa: F401,F403
from data.dataset_readers import *  # noqa: F401,F403
from data.eval_data import *  # noqa: F401,F403,E402
from data.head_reconstruction_data import *  # noqa: F401,F403,E402
