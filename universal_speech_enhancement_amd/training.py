"""Differentiable NCSN++ forward on the HIP operators: the network half of ``ScoreModel.train_step`` with gradients (SURVEY section 8
row f4; reference ``model_wrapper.py:147-208`` driven by ``SGMSEModule.training_step``, ``SGMSE_module.py:46-54``).

Every heavy operator of the network is one ``torch.autograd.Function`` whose forward AND backward are kernels of libuse_hip.so on
NHWC device tensors (fp32 as the reference trains, or 16-bit storage with fp32 parameters: ``compute_dtype``):

* ``conv``      - forward ``use_op_conv_dev`` (implicit-GEMM MFMA kernel with the res-block's epilogue: + Dense_0 row, + shortcut, x 1/sqrt(2);
                  the fp32 parameter tensor is laid out on the device each call, it changes every optimiser step); data gradient = the
                  same kernel on the flipped / transposed weight (w_mode 1); weight and bias gradients ``use_op_wgrad`` (exact-fp32 MFMA
                  for fp32 tensors, 16-bit MFMA for 16-bit ones).  3x3, 1x1 and NIN ([Cin][Cout]) weights.
* ``gn_act``    - ``use_op_gn_act_fwd`` / ``use_op_gn_act_bwd`` (GroupNorm with or without SiLU).
* ``fir``       - ``use_op_fir``; the x2 FIR resamplers are mutual transposes up to the gain.
* ``attn_core`` - ``use_op_attention`` / ``use_op_attention_bwd``.

torch's autograd records the tape and runs the glue between them (channel concatenation, the pyramid sums, the [B, 512] time-embedding
MLP and the Dense_0 projections - library GEMMs on a handful of rows).  The network structure follows the reference's ``NCSNpp.forward``
(``sgmse/backbones/ncsnpp.py:324-501``) for the configuration family of the predict path, with parameters addressed by the reference's
state-dict names.  There is no CPU implementation: without libuse_hip.so / a GPU the first operator raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Sequence

import torch
import torch.nn.functional as F

from . import _lib
from ._lib import UseConvOp, check
from . import training_ops as ops

SQRT1_2 = 0.70710678118654752440


def _conv_dev(x, w, b, w_mode, ntaps, cout, scale=1.0, out_dtype=None, temb=None, res=None):
    """``use_op_conv_dev``: x [B,H,W,Cin] NHWC in fp32 / bf16 / fp16 storage (the MFMA operand type), w / b fp32 DEVICE parameter tensors
    (laid out in x's type on the device); the result in ``out_dtype`` (default: x's)."""
    B, H, W, Cin = x.shape
    out = torch.empty(B, H, W, cout, dtype=out_dtype or x.dtype, device=x.device)
    op = UseConvOp()
    op.B, op.H, op.W, op.C0, op.C1, op.Cout, op.ntaps, op.act, op.variant = B, H, W, Cin, 0, cout, ntaps, 0, 0
    op.dtype, op.out_dtype = ops.dtype_code(x), ops.dtype_code(out)
    op.src0, op.w, op.bias = x.data_ptr(), w.data_ptr(), (b.data_ptr() if b is not None else None)
    op.out_scale, op.out = scale, out.data_ptr()
    op.temb = temb.data_ptr() if temb is not None else None         # fp32 [B][cout], added per item before the scale
    op.res = res.data_ptr() if res is not None else None            # out's type and shape, added before the scale
    lib = _lib.lib()
    n = lib.use_op_conv_dev_workspace(C.byref(op))
    work = torch.empty(n, dtype=torch.uint8, device=x.device)
    check(lib.use_op_conv_dev(C.byref(op), w_mode, C.c_void_p(work.data_ptr()), n, C.c_void_p(torch.cuda.current_stream().cuda_stream)),
          "use_op_conv_dev")
    return out


class _Conv(torch.autograd.Function):
    """y = (conv(x, w) + b [+ temb[b]] [+ res]) * scale - the fused convolution of the res-block (layerspp.py:282-314: Dense_0's row added to
    Conv_0's output, the shortcut added to Conv_1's and the sum scaled by 1/sqrt(2)).  w: [Cout][Cin][3][3], [Cout][Cin][1][1] (reference
    conv3x3 / conv1x1, layerspp.py:31-34) or the NIN matrix [Cin][Cout] (layers.py:639-650).  Channel counts are multiples of 32 (the
    caller zero-pads the 2/4/6-channel ends of the network).  temb: fp32 [B][Cout]; res: the output's shape and type."""

    @staticmethod
    def forward(ctx, x, w, b, out_dtype=None, temb=None, res=None, scale=1.0):
        x, w = x.contiguous(), w.contiguous()
        nin = w.dim() == 2
        ntaps = 1 if nin or w.shape[2] == 1 else 9
        cout = w.shape[1] if nin else w.shape[0]
        ctx.save_for_backward(x, w)
        ctx.nin, ctx.ntaps, ctx.has_bias, ctx.scale = nin, ntaps, b is not None, float(scale)
        ctx.has_temb, ctx.res_dtype = temb is not None, (None if res is None else res.dtype)
        return _conv_dev(x, w, b, 2 if nin else 0, ntaps, cout, scale=scale, out_dtype=out_dtype,
                         temb=None if temb is None else temb.float().contiguous(), res=None if res is None else res.contiguous())

    @staticmethod
    def backward(ctx, gy):
        x, w = ctx.saved_tensors
        g_res = g_temb = None
        if ctx.res_dtype is not None and ctx.needs_input_grad[5]:
            g_res = (gy * ctx.scale).to(ctx.res_dtype)
        if ctx.has_temb and ctx.needs_input_grad[4]:
            g_temb = ops.colsum(gy.contiguous(), scale=ctx.scale)
        # the 2/4/6-channel ends of a 16-bit network change type: the backward runs in the 16-bit one (fp32 input, 16-bit output: the
        # weight gradient takes the input rounded to 16 bits - its products then run at the 16-bit MFMA rate; the input needs no gradient)
        if gy.dtype != x.dtype:
            if x.dtype == torch.float32 and not ctx.needs_input_grad[0]:
                x = x.to(gy.dtype)
            else:
                gy = gy.to(x.dtype)
        gy = gy.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # NIN: dx = dy W^T, i.e. a 1x1 conv whose [cout][cin] weight is W itself; conv: the flipped, transposed weight (w_mode 1)
            gx = _conv_dev(gy, w, None, 0 if ctx.nin else 1, ctx.ntaps, x.shape[3], scale=ctx.scale)
        if ctx.needs_input_grad[1] or (ctx.has_bias and ctx.needs_input_grad[2]):
            dw, db = ops.conv_wgrad(gy, x, ntaps=ctx.ntaps, alpha=ctx.scale, with_bias=ctx.has_bias)
            gw = dw.t().contiguous() if ctx.nin else dw.view(w.shape)
            gb = db
        return gx, gw, gb, None, g_temb, g_res, None


class _GNAct(torch.autograd.Function):
    """act(GroupNorm(groups, eps=1e-6)(x)), act: 0 none, 1 SiLU (layerspp.py:255-257, 286,298)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, groups, act):
        x = x.contiguous()
        y, work = ops.gn_act_fwd(x, gamma, beta, groups, act=act, return_work=True)
        ctx.save_for_backward(x, gamma, beta, work)     # the workspace holds mean / rstd: the backward does not recompute them
        ctx.groups, ctx.act = groups, act
        return y

    @staticmethod
    def backward(ctx, gy):
        x, gamma, beta, work = ctx.saved_tensors
        dx, dg, db = ops.gn_act_bwd(x, gy.contiguous(), gamma, beta, ctx.groups, act=ctx.act, fwd_work=work)
        return dx, dg, db, None, None


class _Fir(torch.autograd.Function):
    """upsample_2d / downsample_2d with the [1,3,3,1] kernel (up_or_down_sampling.py:202-264)."""

    @staticmethod
    def forward(ctx, x, up):
        ctx.up = up
        return ops.fir(x.contiguous(), up=up)

    @staticmethod
    def backward(ctx, gy):
        gy = gy.contiguous()
        return (ops.fir(gy, up=False) * 4.0 if ctx.up else ops.fir(gy, up=True) * 0.25), None


class _AttnCore(torch.autograd.Function):
    """softmax(q k^T / sqrt(C)) v per item, [B,N,C] (AttnBlockpp core, layerspp.py:84-88)."""

    @staticmethod
    def forward(ctx, q, k, v):
        q, k, v = q.contiguous(), k.contiguous(), v.contiguous()
        ctx.save_for_backward(q, k, v)
        return ops.attention_core(q, k, v)

    @staticmethod
    def backward(ctx, gO):
        q, k, v = ctx.saved_tensors                  # the backward kernels are fp32 (N <= 16 tokens at the training shapes)
        dq, dk, dv = ops.attention_core_bwd(q.float(), k.float(), v.float(), gO.float().contiguous())
        return dq.to(q.dtype), dk.to(q.dtype), dv.to(q.dtype)


conv, gn_act, fir, attn_core = _Conv.apply, _GNAct.apply, _Fir.apply, _AttnCore.apply


def _pad_to(t, dim, n):
    """Zero-pad dimension ``dim`` of a parameter to ``n`` entries (differentiable: the gradient is the slice)."""
    if t.shape[dim] == n:
        return t
    pad = [0, 0] * (t.dim() - 1 - dim) + [0, n - t.shape[dim]]
    return F.pad(t, pad)


def _gn(x, P, prefix, act):
    Cc = x.shape[3]
    return gn_act(x, P[prefix + ".weight"], P[prefix + ".bias"], min(Cc // 4, 32), act)


def _resblock(x, temb_act, P, p, up=False, down=False):
    """ResnetBlockBigGANpp.forward (layerspp.py:282-314), skip_rescale, dropout 0."""
    h = _gn(x, P, p + ".GroupNorm_0", 1)
    if up or down:
        h, x = fir(h, up), fir(x, up)
    dense = None if temb_act is None else F.linear(temb_act, P[p + ".Dense_0.weight"], P[p + ".Dense_0.bias"])
    h = conv(h, P[p + ".Conv_0.weight"], P[p + ".Conv_0.bias"], None, dense)               # + Dense_0(act(temb)) per item, in the epilogue
    h = _gn(h, P, p + ".GroupNorm_1", 1)
    if (p + ".Conv_2.weight") in P:
        x = conv(x, P[p + ".Conv_2.weight"], P[p + ".Conv_2.bias"])
    return conv(h, P[p + ".Conv_1.weight"], P[p + ".Conv_1.bias"], None, None, x, SQRT1_2)  # (shortcut + Conv_1(h)) / sqrt(2) in the epilogue


def _attn_block(x, P, p):
    """AttnBlockpp.forward (layerspp.py:77-93), skip_rescale."""
    B, H, W, Cc = x.shape
    h = _gn(x, P, p + ".GroupNorm_0", 0)
    q, k, v = (conv(h, P[f"{p}.NIN_{i}.W"], P[f"{p}.NIN_{i}.b"]).view(B, H * W, Cc) for i in range(3))
    a = attn_core(q, k, v).view(B, H, W, Cc)
    return conv(a, P[p + ".NIN_3.W"], P[p + ".NIN_3.b"], None, None, x, SQRT1_2)


def ncsnpp_forward_train(P: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, ch_mult: Sequence[int], num_res_blocks: int,
                         conditional: bool = True, scale_by_sigma: bool = True, compute_dtype=torch.float32) -> torch.Tensor:
    """``NCSNpp.forward`` (ncsnpp.py:324-501) with a tape.  P: the backbone's parameters under the reference's state-dict names
    (fp32, on the GPU); x complex64 [B, n, F, T'] (n = 2: cat[x_t, Y]; 3: + Y_denoised; 1: discriminative network, conditional = scale_by_sigma = False); t float32 [B].
    ``compute_dtype`` torch.bfloat16 / float16: mixed precision - activations and their gradients stored in 16 bits, convolutions (forward
    and data gradient) on the 16-bit MFMA kernels with the fp32 parameters laid out in that type per call; parameters, parameter
    gradients (exact-fp32 MFMA contraction of the 16-bit tensors), GroupNorm statistics, the time embedding, the 2/4/6-channel input and
    output pyramids and the loss stay fp32.  Returns complex64 [B, 1, F, T']."""
    cd = compute_dtype
    if not x.is_cuda:
        from .hip_engine import UseHipError
        raise UseHipError("NCSN++ (HIP) needs CUDA/ROCm tensors: the network has no CPU implementation")
    L = len(ch_mult)
    B, nin, Fq, T = x.shape
    if Fq % (1 << (L - 1)) or T % (1 << (L - 1)):
        raise ValueError(f"the {L - 1} resampling stages need F and T' to be multiples of {1 << (L - 1)} (got {Fq} x {T})")
    # ncsnpp.py:333-347: channels (x.re, x.im, y.re, y.im, ...) -> NHWC, zero-padded to 32 channels; :372-374 (not centered): 2 x - 1
    x4 = torch.view_as_real(x.detach()).permute(0, 2, 3, 1, 4).reshape(B, Fq, T, 2 * nin).float()
    x4 = F.pad(2.0 * x4 - 1.0, (0, 32 - 2 * nin)).contiguous()
    if not conditional:                                                                        # ncsnpp.py:352,364-370
        temb_act, m = None, 1
    else:
        # GaussianFourierProjection(log t) (layerspp.py:37-39; W is not trained there) -> Linear -> SiLU -> Linear (ncsnpp.py:351-368)
        xp = torch.log(t.float())[:, None] * P["all_modules.0.W"].detach()[None, :] * (2 * math.pi)
        e = torch.cat([torch.sin(xp), torch.cos(xp)], dim=-1)
        e = F.linear(e, P["all_modules.1.weight"], P["all_modules.1.bias"])
        temb_act = F.silu(F.linear(F.silu(e), P["all_modules.2.weight"], P["all_modules.2.bias"]))   # every consumer takes SiLU(temb)
        m = 3
    pyr_in = x4
    hs = [conv(x4, _pad_to(P[f"all_modules.{m}.weight"], 1, 32), P[f"all_modules.{m}.bias"], cd)]
    m += 1
    for lvl in range(L):
        for _ in range(num_res_blocks):
            hs.append(_resblock(hs[-1], temb_act, P, f"all_modules.{m}")); m += 1
        if lvl != L - 1:
            h = _resblock(hs[-1], temb_act, P, f"all_modules.{m}", down=True); m += 1
            pyr_in = fir(pyr_in, False)                                                         # ncsnpp.py:404
            h = conv(pyr_in, _pad_to(P[f"all_modules.{m}.Conv_0.weight"], 1, 32), P[f"all_modules.{m}.Conv_0.bias"], cd) + h   # Combine 'sum'
            m += 1
            hs.append(h)
    h = _resblock(hs[-1], temb_act, P, f"all_modules.{m}"); m += 1
    h = _attn_block(h, P, f"all_modules.{m}"); m += 1
    h = _resblock(h, temb_act, P, f"all_modules.{m}"); m += 1
    pyramid = None
    for lvl in reversed(range(L)):
        for _ in range(num_res_blocks + 1):
            h = _resblock(torch.cat([h, hs.pop()], dim=3), temb_act, P, f"all_modules.{m}"); m += 1
        ph = _gn(h, P, f"all_modules.{m}", 1); m += 1                                           # ncsnpp.py:443,457
        ph = conv(ph, _pad_to(P[f"all_modules.{m}.weight"], 0, 32), _pad_to(P[f"all_modules.{m}.bias"], 0, 32), torch.float32); m += 1
        pyramid = ph if pyramid is None else fir(pyramid, True) + ph                            # ncsnpp.py:456-461
        if lvl != 0:
            h = _resblock(h, temb_act, P, f"all_modules.{m}", up=True); m += 1
    assert not hs
    if scale_by_sigma:
        pyramid = pyramid / t.float()[:, None, None, None]                                      # ncsnpp.py:492-494
    wo = _pad_to(_pad_to(P["output_layer.weight"], 1, 32), 0, 32)
    out = conv(pyramid, wo, _pad_to(P["output_layer.bias"], 0, 32))[..., :2]                    # ncsnpp.py:497
    return torch.view_as_complex(out.contiguous()).unsqueeze(1)                                 # ncsnpp.py:498-500
