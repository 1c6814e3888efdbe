"""Parameter grouping of the reference (optim/misc.py:14-50): no weight decay for names containing
'bias' / 'LayerNorm.bias' / 'LayerNorm.weight'; task-head parameters (outside `v_encoder`) get
lr_mul * learning_rate."""
from .adamw import AdamW

NO_DECAY = ("bias", "LayerNorm.bias", "LayerNorm.weight")


def build_optimizer(model, opts):
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    enc = [(n, p) for n, p in named if "v_encoder" in n]
    top = [(n, p) for n, p in named if "v_encoder" not in n]

    def split(items, decay):
        return [p for n, p in items if any(t in n for t in NO_DECAY) != decay]

    groups = [
        {"params": split(top, True), "lr": opts.lr_mul * opts.learning_rate, "weight_decay": opts.weight_decay},
        {"params": split(top, False), "lr": opts.lr_mul * opts.learning_rate, "weight_decay": 0.0},
        {"params": split(enc, True), "weight_decay": opts.weight_decay},
        {"params": split(enc, False), "weight_decay": 0.0},
    ]
    if opts.optim != "adamw":
        raise ValueError("hero_amd ships the fused AdamW only (reference default for every HERO config)")
    return AdamW(groups, lr=opts.learning_rate, betas=tuple(opts.betas))
