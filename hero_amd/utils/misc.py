"""Small helpers with the reference's behaviour (utils/misc.py)."""
import random

import numpy as np
import torch


class NoOp(object):
    """Swallow any method call (used on non-zero ranks)."""

    def __getattr__(self, name):
        return self.noop

    def noop(self, *args, **kwargs):
        return None


def set_dropout(model, drop_p):
    """Rewrite `.p` of every nn.Dropout child (utils/misc.py:32-38). The HIP kernels read `.p` at
    call time, so this takes effect on the next forward."""
    for _, module in model.named_modules():
        if isinstance(module, torch.nn.Dropout) and module.p != drop_p:
            module.p = drop_p


def set_random_seed(seed):
    from .. import functional as HF
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)
    HF.manual_seed(seed)
