"""ste_round (compressai/ops/ops.py:18-31): round with a straight-through gradient.  Not used by
HESIC; kept because the name is part of the reference's public surface."""
import torch


def ste_round(x):
    return torch.round(x) - x.detach() + x
