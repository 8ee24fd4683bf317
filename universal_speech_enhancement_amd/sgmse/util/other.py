"""``pad_spec`` of the reference (``sgmse/util/other.py:128-135``): zero-pad the frame axis on the right
to the next multiple of 64 so the 6 FIR down/up-samplings of NCSN++ round-trip."""
import torch


def pad_spec(Y: torch.Tensor) -> torch.Tensor:
    T = Y.size(3)
    num_pad = (64 - T % 64) % 64
    return torch.nn.functional.pad(Y, (0, num_pad, 0, 0))
