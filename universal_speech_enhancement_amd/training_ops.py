"""Backward pass of one ``ResnetBlockBigGANpp`` on the HIP operators (minimum slice): the gradient half of the
reference's ``ScoreModel.train_step`` (model_wrapper.py:147-208, driven by ``SGMSEModule.training_step``, SGMSE_module.py:46-54)
needs, per res-block, the data and weight gradients of two 3x3 convolutions and the 1x1 shortcut, the backward of two
GroupNorm + SiLU pairs and of Dense_0(SiLU(temb)), and - in the up / down blocks - the transposes of the FIR resamplers (which are
each other up to the gain: ``use_op_fir`` again).  fp32 storage, NHWC device tensors.

* data gradient of a convolution = the forward implicit-GEMM kernel (``use_op_conv``) on the flipped, transposed weights;
* weight gradient = ``use_op_wgrad`` (pixels as the K of an exact-fp32 MFMA contraction);
* GroupNorm + SiLU backward = ``use_op_gn_act_bwd``; Dense_0 = ``use_op_colsum`` + ``use_op_dense_bwd``.

The forward activations the backward needs (block input, Conv_0 output) are recomputed / taken from the forward operators by the caller;
this module only moves pointers.  The whole-network tape built on these operators is ``training.py``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import UseConvOp, check

SQRT1_2 = 0.70710678118654752440


def _p(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _pad32(c):
    return (c + 31) // 32 * 32


_DT = {torch.float32: 0, torch.bfloat16: 1, torch.float16: 2}


def dtype_code(t):
    """The library's storage-type code of a tensor (0 fp32, 1 bf16, 2 fp16)."""
    return _DT[t.dtype]


def conv_fwd(x, w, bias=None, coef=None, act=0, temb=None, res=None, x0=None, w2=None, scale=1.0, ntaps=9, stats=None):
    """The library's fused convolution on fp32 NHWC tensors whose channel counts are multiples of 32 (``use_op_conv``)."""
    B, H, W, Cin = x.shape
    Cout = w.shape[0]
    out = torch.empty(B, H, W, Cout, dtype=torch.float32, device=x.device)
    op = UseConvOp()
    op.B, op.H, op.W, op.C0, op.C1, op.Cout, op.ntaps, op.act, op.dtype, op.out_dtype, op.variant = B, H, W, Cin, 0, Cout, ntaps, act, 0, 0, 0
    op.src0 = x.data_ptr()
    keep = [np.ascontiguousarray(w, dtype=np.float32)]
    op.w = keep[0].ctypes.data
    if bias is not None:
        keep.append(np.ascontiguousarray(bias, dtype=np.float32)); op.bias = keep[-1].ctypes.data
    if x0 is not None:
        op.XC0 = x0.shape[3]; op.x0 = x0.data_ptr()
        keep.append(np.ascontiguousarray(w2, dtype=np.float32)); op.w2 = keep[-1].ctypes.data
    op.coef = coef.data_ptr() if coef is not None else None
    op.temb = temb.data_ptr() if temb is not None else None
    op.res = res.data_ptr() if res is not None else None
    op.out_scale = scale; op.out = out.data_ptr(); op.stats = stats.data_ptr() if stats is not None else None
    check(_lib.lib().use_op_conv(C.byref(op), _stream()), "use_op_conv")
    return out


def conv_dgrad(dy, w, scale=1.0):
    """dX of y = conv3x3(x, w) (or conv1x1 for a 2-D w): the same kernel on w'[ci][co][ky][kx] = w[co][ci][2-ky][2-kx]."""
    if w.ndim == 4:
        wt = np.ascontiguousarray(np.flip(np.asarray(w), (2, 3)).transpose(1, 0, 2, 3))
        return conv_fwd(dy, wt, scale=scale)
    return conv_fwd(dy, np.ascontiguousarray(np.asarray(w).T), scale=scale, ntaps=1)


def conv_wgrad(dy, x, ntaps=9, alpha=1.0, with_bias=True, tiled=True):
    """dW [Cout][Cin][3][3] (or [Cout][Cin]) and db (fp32) of y = conv(x, w) + b from dy (``use_op_wgrad``); dy and x share a storage
    type (fp32 / bf16 / fp16; the contraction is exact-fp32 MFMA either way).  ``tiled=False``: the small-tile kernel with atomic
    accumulation (fp32 only)."""
    B, H, W, Cout = dy.shape
    Cin = x.shape[3]
    assert dy.dtype == x.dtype
    dw = torch.empty(Cout, Cin, *((3, 3) if ntaps == 9 else ()), dtype=torch.float32, device=dy.device)
    db = torch.empty(Cout, dtype=torch.float32, device=dy.device) if with_bias else None
    lib = _lib.lib()
    n = lib.use_op_wgrad_workspace(B, H, W, Cout, Cin, ntaps, dtype_code(x)) if tiled else 0
    work = torch.empty(n, dtype=torch.float32, device=dy.device) if n else None
    check(lib.use_op_wgrad(_p(dy), _p(x), dtype_code(x), _p(dw), _p(db), B, H, W, Cout, Cin, ntaps, alpha, _p(work), n, _stream()), "use_op_wgrad")
    return dw, db


def gn_act_bwd(x, dy, gamma, beta, groups, act=1, add=None, add_scale=1.0, eps=1e-6, fwd_work=None):
    """``fwd_work``: the workspace ``gn_act_fwd(..., return_work=True)`` ran in for the same x - its statistics are reused."""
    B, H, W, Cc = x.shape
    assert dy.dtype == x.dtype and (add is None or add.dtype == x.dtype)
    work = fwd_work if fwd_work is not None else torch.empty(_lib.lib().use_op_gn_workspace(B, Cc, groups), dtype=torch.float32, device=x.device)
    dx = torch.empty_like(x)
    dg, dbt = torch.empty(Cc, device=x.device), torch.empty(Cc, device=x.device)
    check(_lib.lib().use_op_gn_act_bwd(_p(x), _p(dy), dtype_code(x), _p(gamma), _p(beta), groups, eps, act, _p(add), add_scale, B, H * W, Cc,
                                        _p(work), int(fwd_work is not None), _p(dx), _p(dg), _p(dbt), _stream()), "use_op_gn_act_bwd")
    return dx, dg, dbt


def gn_act_fwd(x, gamma, beta, groups, act=1, eps=1e-6, return_work=False):
    """act(GroupNorm(x)) - the operand of the following convolution's weight gradient, recomputed from the stored pre-activation."""
    B, H, W, Cc = x.shape
    work = torch.empty(_lib.lib().use_op_gn_workspace(B, Cc, groups), dtype=torch.float32, device=x.device)
    y = torch.empty_like(x)
    check(_lib.lib().use_op_gn_act_fwd(_p(x), dtype_code(x), _p(gamma), _p(beta), groups, eps, act, B, H * W, Cc, _p(work), _p(y), _stream()),
          "use_op_gn_act_fwd")
    return (y, work) if return_work else y


def fir(x, up):
    """upsample_2d / downsample_2d of an NHWC tensor (``use_op_fir``; fp32 / bf16 / fp16 storage).  The two are mutual transposes up to the
    gain: backward(upsample_2d)(g) = 4 downsample_2d(g), backward(downsample_2d)(g) = upsample_2d(g) / 4."""
    B, H, W, Cc = x.shape
    H2, W2 = (H * 2, W * 2) if up else (H // 2, W // 2)
    out = torch.empty(B, H2, W2, Cc, dtype=x.dtype, device=x.device)
    check(_lib.lib().use_op_fir(_p(x), dtype_code(x), None, 0, None, _p(out), B, H, W, Cc, int(up), _stream()), "use_op_fir")
    return out


def dense_bwd(g, temb, Wd):
    B, Cout = g.shape
    K = temb.shape[1]
    dW, db, dt = torch.empty(Cout, K, device=g.device), torch.empty(Cout, device=g.device), torch.empty(B, K, device=g.device)
    check(_lib.lib().use_op_dense_bwd(_p(g), _p(temb), _p(Wd), B, K, Cout, _p(dW), _p(db), _p(dt), _stream()), "use_op_dense_bwd")
    return dW, db, dt


def colsum(x, scale=1.0):
    """out[b][c] = scale * sum over the pixels of x[b, :, :, c] (fp32; ``use_op_colsum``, pixel-sliced kernel)."""
    B, H, W, Cc = x.shape
    out = torch.empty(B, Cc, dtype=torch.float32, device=x.device)
    work = torch.empty(128 * B * Cc, dtype=torch.float32, device=x.device)
    check(_lib.lib().use_op_colsum(_p(x), dtype_code(x), B, H * W, Cc, scale, _p(out), _p(work), _stream()), "use_op_colsum")
    return out


def resblock_backward(x, h1, temb, gy, W, groups0, groups1, up=False, down=False):
    """Gradients of y = (shortcut(r(x)) + Conv_1(SiLU(GN_1(h1)))) / sqrt(2), h1 = Conv_0(r(SiLU(GN_0(x)))) + Dense_0(SiLU(temb)), given
    gy = dL/dy; r = identity, or the FIR x2 up / down resampling of the BigGAN block (layerspp.py:286-300).
    x [B,H,W,Cin], h1 and gy at the block's output resolution [B,H',W',Cout]: fp32 NHWC on the GPU, channels multiples of 32
    (zero-padded; `groups*` count the padding's all-zero groups as well).  W: the block's parameters (numpy conv weights, padded alike;
    GroupNorm / Dense parameters as CUDA tensors under the key + '.dev').  Returns dict of gradients."""
    g = {}
    resample = up or down
    back = (lambda t: fir(t, up=False) * 4.0) if up else (lambda t: fir(t, up=True) * 0.25) if down else (lambda t: t)   # r^T
    a1 = gn_act_fwd(h1, W["GroupNorm_1.weight.dev"], W["GroupNorm_1.bias.dev"], groups1)     # operand of Conv_1's weight gradient (recomputed)
    g["Conv_1.weight"], g["Conv_1.bias"] = conv_wgrad(gy, a1, alpha=SQRT1_2)
    da1 = conv_dgrad(gy, W["Conv_1.weight"], scale=SQRT1_2)
    dh1, g["GroupNorm_1.weight"], g["GroupNorm_1.bias"] = gn_act_bwd(h1, da1, W["GroupNorm_1.weight.dev"], W["GroupNorm_1.bias.dev"], groups1)
    # Dense_0(SiLU(temb)) is broadcast over the pixels of h1
    g["Dense_0.weight"], g["Dense_0.bias"], g["temb"] = dense_bwd(colsum(dh1), temb, W["Dense_0.weight.dev"])
    a0 = gn_act_fwd(x, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], groups0)
    a0r = fir(a0, up=up) if resample else a0                              # Conv_0 sees the resampled activation
    g["Conv_0.weight"], g["Conv_0.bias"] = conv_wgrad(dh1, a0r)
    da0 = back(conv_dgrad(dh1, W["Conv_0.weight"]))                       # back through the resampler to the input resolution
    if "Conv_2.weight" in W:                                              # 1x1 shortcut on the (resampled) raw input
        xr = fir(x, up=up) if resample else x
        g["Conv_2.weight"], g["Conv_2.bias"] = conv_wgrad(gy, xr, ntaps=1, alpha=SQRT1_2)
        dsc = back(conv_dgrad(gy, W["Conv_2.weight"], scale=SQRT1_2))
        g["x"], g["GroupNorm_0.weight"], g["GroupNorm_0.bias"] = gn_act_bwd(x, da0, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], groups0,
                                                                           add=dsc, add_scale=1.0)
    else:                                                                 # identity shortcut
        g["x"], g["GroupNorm_0.weight"], g["GroupNorm_0.bias"] = gn_act_bwd(x, da0, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], groups0,
                                                                           add=gy, add_scale=SQRT1_2)
    return g


def attention_core(q, k, v):
    """softmax(q k^T / sqrt(C)) v on [B,N,C] tensors (``use_op_attention``; fp32 / bf16 / fp16 storage)."""
    B, N, Cc = q.shape
    out = torch.empty_like(q)
    check(_lib.lib().use_op_attention(_p(q), _p(k), _p(v), _p(out), dtype_code(q), B, N, Cc, _stream()), "use_op_attention")
    return out


def attention_core_bwd(q, k, v, dO):
    B, N, Cc = q.shape
    work = torch.empty(2 * B * N * N, dtype=torch.float32, device=q.device)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(q)
    check(_lib.lib().use_op_attention_bwd(_p(q), _p(k), _p(v), _p(dO), _p(work), _p(dq), _p(dk), _p(dv), B, N, Cc, _stream()), "use_op_attention_bwd")
    return dq, dk, dv


def attn_block_backward(x, gy, W, groups):
    """Gradients of AttnBlockpp (layerspp.py:60-93): y = (x + NIN_3(softmax(q k^T / sqrt(C)) v)) / sqrt(2), q, k, v = NIN_0..2(GroupNorm(x)).
    x, gy [B,H,W,C] fp32 NHWC on the GPU (C a multiple of 32); W: 'NIN_i.W' numpy [C][C] (the reference's [cin][cout]) and 'NIN_i.b',
    GroupNorm parameters as CUDA tensors under key + '.dev'.  The forward activations (h, q, k, v, attention output) are recomputed
    with the forward operators.  NIN = 1x1 convolution with weight W^T."""
    B, H, Wd, Cc = x.shape
    N = H * Wd
    g = {}
    h = gn_act_fwd(x, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], groups, act=0)
    wt = [np.ascontiguousarray(np.asarray(W[f"NIN_{i}.W"]).T) for i in range(4)]            # conv weights [cout][cin]
    q, k, v = (conv_fwd(h, wt[i], bias=W[f"NIN_{i}.b"], ntaps=1) for i in range(3))
    a = attention_core(q.view(B, N, Cc), k.view(B, N, Cc), v.view(B, N, Cc)).view(B, H, Wd, Cc)
    # y = (x + NIN_3(a)) / sqrt(2)
    dw3, g["NIN_3.b"] = conv_wgrad(gy, a, ntaps=1, alpha=SQRT1_2)
    g["NIN_3.W"] = dw3.t().contiguous()                                                   # back to the reference's [cin][cout]
    da = conv_fwd(gy, np.ascontiguousarray(np.asarray(W["NIN_3.W"])), scale=SQRT1_2, ntaps=1)    # dgrad of a 1x1: the transposed matrix
    dq, dk, dv = attention_core_bwd(q.view(B, N, Cc), k.view(B, N, Cc), v.view(B, N, Cc), da.view(B, N, Cc))
    dh = None
    for i, d in enumerate((dq, dk, dv)):
        d4 = d.view(B, H, Wd, Cc)
        dwi, g[f"NIN_{i}.b"] = conv_wgrad(d4, h, ntaps=1)
        g[f"NIN_{i}.W"] = dwi.t().contiguous()
        di = conv_fwd(d4, np.ascontiguousarray(np.asarray(W[f"NIN_{i}.W"])), ntaps=1)
        dh = di if dh is None else dh + di
    g["x"], g["GroupNorm_0.weight"], g["GroupNorm_0.bias"] = gn_act_bwd(x, dh, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], groups, act=0,
                                                                       add=gy, add_scale=SQRT1_2)
    return g
