"""Weight-loading helpers with the reference's semantics (model/modeling_utils.py)."""
import logging

import torch

logger = logging.getLogger(__name__)


def mask_logits(target, mask, eps=-1e4):
    """Multiplicative + additive masking used by the heads (modeling_utils.py:42-43)."""
    return target * mask + (1 - mask) * eps


def pad_tensor_to_mul(tensor, dim=0, mul=8):
    """Zero-pad `dim` up to a multiple of `mul`; returns (tensor, n_pad)."""
    n_pad = (-tensor.size(dim)) % mul
    if n_pad == 0:
        return tensor, 0
    shape = list(tensor.size())
    shape[dim] = n_pad
    return torch.cat([tensor, tensor.new_zeros(shape)], dim=dim), n_pad


def load_partial_checkpoint(checkpoint, n_layers, skip_layers=True):
    """Keep every (12 / n_layers)-th RoBERTa layer, renumbered 0..n_layers-1
    (modeling_utils.py:46-65: layers {1,3,5,7,9,11} -> {0..5} for n_layers = 6)."""
    if not skip_layers:
        return checkpoint
    gap = 12 // n_layers
    keep = {str(src): str(dst) for dst, src in enumerate(range(gap - 1, 12, gap))}
    marker = "roberta.encoder.layer."
    out = {}
    for key, val in checkpoint.items():
        if marker not in key:
            out[key] = val
            continue
        parts = key.split(".")
        if parts[3] in keep:
            parts[3] = keep[parts[3]]
            out[".".join(parts)] = val
    return out


def load_pretrained_weight(model, state_dict):
    """Non-strict load with the TF-style gamma/beta renames and an optional 'roberta.' prefix
    (modeling_utils.py:68-121)."""
    sd = {}
    for key, val in state_dict.items():
        sd[key.replace("gamma", "weight").replace("beta", "bias")] = val
    prefix = ""
    if not hasattr(model, "roberta") and any(k.startswith("roberta.") for k in sd):
        prefix = "roberta."
    if prefix:
        sd = {k[len(prefix):]: v for k, v in sd.items() if k.startswith(prefix)}
    result = model.load_state_dict(sd, strict=False)
    if result.missing_keys:
        logger.info("Weights of %s not initialized from pretrained model: %s",
                    model.__class__.__name__, result.missing_keys)
    if result.unexpected_keys:
        logger.info("Weights from pretrained model not used in %s: %s",
                    model.__class__.__name__, result.unexpected_keys)
    return model
