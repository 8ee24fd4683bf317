"""State-dict layout (ordered key -> shape) of the NCSN++ configurations served by the HIP path; mirrors the module
construction order of the reference ``sgmse/backbones/ncsnpp.py:116-316``."""
from __future__ import annotations

from collections import OrderedDict
from typing import Sequence, Tuple

# ----------------------------------------------------------------------------------------------
# Architecture description (mirrors reference ncsnpp.py:42-316 construction order)
# ----------------------------------------------------------------------------------------------

def ncsnpp_param_shapes(
    nf: int = 128,
    ch_mult: Sequence[int] = (1, 1, 2, 2, 2, 2, 2),
    num_res_blocks: int = 2,
    input_channels: int = 4,
    conditional: bool = True,
) -> "OrderedDict[str, Tuple[int, ...]]":
    """Ordered {state-dict key: shape} for the configuration family used by the predict path:
    biggan res-blocks, FIR resampling, progressive output_skip / input_skip with 'sum' combine,
    fourier embedding, attention only at the bottleneck (reference ncsnpp.py:42-69 defaults).
    """
    shapes: "OrderedDict[str, Tuple[int, ...]]" = OrderedDict()
    shapes["output_layer.weight"] = (2, input_channels, 1, 1)
    shapes["output_layer.bias"] = (2,)
    idx = 0

    def add_plain(suffix, shape):
        shapes[f"all_modules.{idx}.{suffix}"] = tuple(shape)

    temb_dim = nf * 4
    # 0: GaussianFourierProjection
    add_plain("W", (nf,))
    idx += 1
    if conditional:
        add_plain("weight", (temb_dim, 2 * nf)); add_plain("bias", (temb_dim,)); idx += 1
        add_plain("weight", (temb_dim, temb_dim)); add_plain("bias", (temb_dim,)); idx += 1
    # input conv
    add_plain("weight", (nf, input_channels, 3, 3)); add_plain("bias", (nf,)); idx += 1

    def resblock(in_ch, out_ch, resample):
        nonlocal idx
        add_plain("GroupNorm_0.weight", (in_ch,)); add_plain("GroupNorm_0.bias", (in_ch,))
        add_plain("Conv_0.weight", (out_ch, in_ch, 3, 3)); add_plain("Conv_0.bias", (out_ch,))
        # Dense_0 exists even when the network is unconditional (temb_dim is always passed, ncsnpp.py:160-171)
        add_plain("Dense_0.weight", (out_ch, temb_dim)); add_plain("Dense_0.bias", (out_ch,))
        add_plain("GroupNorm_1.weight", (out_ch,)); add_plain("GroupNorm_1.bias", (out_ch,))
        add_plain("Conv_1.weight", (out_ch, out_ch, 3, 3)); add_plain("Conv_1.bias", (out_ch,))
        if in_ch != out_ch or resample:
            add_plain("Conv_2.weight", (out_ch, in_ch, 1, 1)); add_plain("Conv_2.bias", (out_ch,))
        idx += 1

    L = len(ch_mult)
    hs_c = [nf]
    in_ch = nf
    for lvl in range(L):
        for _ in range(num_res_blocks):
            out_ch = nf * ch_mult[lvl]
            resblock(in_ch, out_ch, False)
            in_ch = out_ch
            hs_c.append(in_ch)
        if lvl != L - 1:
            resblock(in_ch, in_ch, True)  # down
            add_plain("Conv_0.weight", (in_ch, input_channels, 1, 1)); add_plain("Conv_0.bias", (in_ch,)); idx += 1
            hs_c.append(in_ch)
    in_ch = hs_c[-1]
    resblock(in_ch, in_ch, False)
    # attention
    add_plain("GroupNorm_0.weight", (in_ch,)); add_plain("GroupNorm_0.bias", (in_ch,))
    for k in range(4):
        add_plain(f"NIN_{k}.W", (in_ch, in_ch)); add_plain(f"NIN_{k}.b", (in_ch,))
    idx += 1
    resblock(in_ch, in_ch, False)
    for lvl in reversed(range(L)):
        for _ in range(num_res_blocks + 1):
            out_ch = nf * ch_mult[lvl]
            resblock(in_ch + hs_c.pop(), out_ch, False)
            in_ch = out_ch
        add_plain("weight", (in_ch,)); add_plain("bias", (in_ch,)); idx += 1       # GroupNorm
        add_plain("weight", (input_channels, in_ch, 3, 3)); add_plain("bias", (input_channels,)); idx += 1
        if lvl != 0:
            resblock(in_ch, in_ch, True)  # up
    assert not hs_c
    return shapes
