"""``batch_broadcast`` (reference ``sgmse/util/tensors.py:4-20``)."""
import torch


def batch_broadcast(a: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """View the per-batch vector ``a`` so that it broadcasts over all non-batch dims of ``x``."""
    if a.dim() != 1:
        a = a.squeeze()
        if a.dim() != 1:
            raise ValueError(f"Don't know how to batch-broadcast tensor `a` with more than one effective dimension (shape {a.shape})")
    if a.shape[0] != x.shape[0] and a.shape[0] != 1:
        raise ValueError(f"Don't know how to batch-broadcast shape {a.shape} over {x.shape} as the batch dimension is not matching")
    return a.view(x.shape[0], *([1] * (x.dim() - 1)))
