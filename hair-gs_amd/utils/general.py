"""Small numeric helpers of the training loop (counterparts of the reference's utils/general.py:22-124)."""
import os
import random

import numpy as np
import torch


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """Log-linear interpolation lr_init -> lr_final over max_steps, optional sine warm-up (reference :35-68).
    The reference never passes lr_delay_steps, so the warm-up is inactive in practice."""
    # (the two logarithms are the same numbers at every step: taken once; the clip of a float to [0, 1] is min / max -- the
    # training loop calls three of these per iteration, and four numpy scalar calls each were 2 % of its wall clock)
    zero = lr_init == 0.0 and lr_final == 0.0
    log_init, log_final = (0.0, 0.0) if zero else (np.log(lr_init), np.log(lr_final))

    def schedule(step):
        if step < 0 or zero:
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * min(max(step / lr_delay_steps, 0), 1))
        t = min(max(step / max_steps, 0), 1)
        return delay * np.exp(log_init * (1 - t) + log_final * t)
    return schedule


def strip_lowerdiag(L):
    """[N,3,3] symmetric -> [N,6] (xx, xy, xz, yy, yz, zz): the cov3D_precomp layout of the rasterizer."""
    return torch.stack([L[:, 0, 0], L[:, 0, 1], L[:, 0, 2], L[:, 1, 1], L[:, 1, 2], L[:, 2, 2]], dim=1)


def strip_symmetric(sym):
    return strip_lowerdiag(sym)


def safe_state(silent=True, device=None):
    """Seeds as the reference's safe_state (utils/general.py:113-116)."""
    random.seed(0)
    np.random.seed(0)
    torch.manual_seed(0)
    if device is not None and torch.cuda.is_available():
        torch.cuda.set_device(device)


def enable_accelerated_rasterization():
    """The reference selects its DISTWAR backward through env vars (utils/general.py:119-124).  libhgs.so has one
    backward (wave64 reduction + deterministic gather); the variables are set for scripts that read them."""
    os.environ["BW_IMPLEMENTATION"] = "1"
    os.environ["BALANCE_THRESHOLD"] = "8"
