from .adamw import AdamW  # noqa: F401
from .misc import build_optimizer  # noqa: F401
from .sched import get_lr_sched, warmup_linear  # noqa: F401
