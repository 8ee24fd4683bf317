"""Per-operator GPU parity: the HIP kernels one at a time (C ABI: use_op_fir / use_op_conv / use_op_gn_finalize / use_op_attention)
against the per-operator golden vectors generated from the reference itself (oracle/gen_golden.py): upsample_2d / downsample_2d
(fir.npz), ResnetBlockBigGANpp in its five shapes (resblock_{plain,widen,down,up,cat}.npz) and AttnBlockpp (attn.npz).

The test composes a res-block from the operators exactly as the engine does (use_engine.cpp, Fwd::resblock): FIR resampling with the
GroupNorm+SiLU of the block input fused, Conv_0 with bias + Dense_0(temb), GroupNorm_1 from the totals Conv_0 accumulated, Conv_1
with the fused 1x1 shortcut (or the residual) and the 1/sqrt(2).  The goldens use 16 / 32 / 48 channels; the kernels take
multiples of 32, so tensors are zero-padded (zero channels with zero weights change nothing).

Tolerances, relative to the golden tensor's max: fp32 storage ~1e-6 (accumulation order); 16-bit storage: 4 units in the last place of
the storage type (bf16 4 x 2^-8 = 1.6e-2, fp16 4 x 2^-11 = 2.0e-3: two roundings of the operands, one of the result, tests/lowprec.py)
- a property of the number format, not of this build's last run.  Each test prints what it measured."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

import lowprec as lp
from universal_speech_enhancement_amd import _lib
from universal_speech_enhancement_amd._lib import UseConvOp, check

pytestmark = pytest.mark.gpu
TD = {0: torch.float32, 1: torch.bfloat16, 2: torch.float16}
SQRT2 = 2.0 ** 0.5


def _pad32(c):
    return (c + 31) // 32 * 32


def _nhwc(x, cpad, dt):
    """[B,C,H,W] fp32 (CPU) -> NHWC, channels zero-padded to cpad, storage dtype, on the GPU."""
    B, Cc, H, W = x.shape
    y = torch.zeros(B, H, W, cpad, dtype=torch.float32)
    y[..., :Cc] = x.permute(0, 2, 3, 1)
    return y.to(TD[dt]).cuda().contiguous()


def _nchw(y, c):
    return y.float().cpu()[..., :c].permute(0, 3, 1, 2).contiguous()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _gn_coef(x, gamma, beta, cpad):
    """GroupNorm(min(C//4, 32), C, eps=1e-6) of x [B,C,H,W] folded to (a, b) per (item, channel): [B][cpad][2] on the GPU."""
    B, Cc = x.shape[:2]
    G = min(Cc // 4, 32)
    xg = x.reshape(B, G, -1).double()
    mean, var = xg.mean(-1), xg.var(-1, unbiased=False)
    rstd = 1.0 / torch.sqrt(var + 1e-6)
    cpg = Cc // G
    a = gamma[None, :].double() * rstd.repeat_interleave(cpg, 1)
    b = beta[None, :].double() - mean.repeat_interleave(cpg, 1) * a
    coef = torch.zeros(B, cpad, 2)
    coef[:, :Cc, 0] = a.float(); coef[:, :Cc, 1] = b.float()
    return coef.cuda().contiguous()


def _conv(dt, B, H, W, srcs, coef, act, w, bias, temb, xs, w2, res, scale, cout, stats, variant=0, ntaps=9, out_dt=None):
    """srcs / xs: lists of (device tensor NHWC, channels as stored); w: numpy [cout][sum stored cin][ntaps...]; returns NHWC out."""
    cop = _pad32(cout)
    out_dt = dt if out_dt is None else out_dt
    out = torch.empty(B, H, W, cop, dtype=TD[out_dt], device="cuda")
    op = UseConvOp()
    op.B, op.H, op.W, op.Cout, op.ntaps, op.act, op.dtype, op.out_dtype, op.variant = B, H, W, cop, ntaps, act, dt, out_dt, variant
    op.C0 = srcs[0][1]; op.src0 = srcs[0][0].data_ptr()
    op.C1 = srcs[1][1] if len(srcs) > 1 else 0; op.src1 = srcs[1][0].data_ptr() if len(srcs) > 1 else None
    op.XC0 = xs[0][1] if xs else 0; op.x0 = xs[0][0].data_ptr() if xs else None
    op.XC1 = xs[1][1] if xs and len(xs) > 1 else 0; op.x1 = xs[1][0].data_ptr() if xs and len(xs) > 1 else None
    keep = [np.ascontiguousarray(w, dtype=np.float32)]
    op.w = keep[0].ctypes.data
    if bias is not None:
        keep.append(np.ascontiguousarray(bias, dtype=np.float32)); op.bias = keep[-1].ctypes.data
    if w2 is not None:
        keep.append(np.ascontiguousarray(w2, dtype=np.float32)); op.w2 = keep[-1].ctypes.data
    op.coef = coef.data_ptr() if coef is not None else None
    op.temb = temb.data_ptr() if temb is not None else None
    op.res = res.data_ptr() if res is not None else None
    op.out_scale = scale; op.out = out.data_ptr(); op.stats = stats.data_ptr() if stats is not None else None
    check(_lib.lib().use_op_conv(C.byref(op), _stream()), "use_op_conv")
    return out


def _fir(x, dt, coef, act, up):
    B, H, W, Cc = x.shape
    H2, W2 = (H * 2, W * 2) if up else (H // 2, W // 2)
    oa = torch.empty(B, H2, W2, Cc, dtype=x.dtype, device="cuda") if coef is not None else None
    orw = torch.empty(B, H2, W2, Cc, dtype=x.dtype, device="cuda")
    check(_lib.lib().use_op_fir(_ptr(x), dt, _ptr(coef), act, _ptr(oa), _ptr(orw), B, H, W, Cc, int(up), _stream()), "use_op_fir")
    torch.cuda.synchronize()
    return oa, orw


def _rel(a, b):
    return float((a - b).abs().max() / b.abs().max())


@pytest.mark.parametrize("dt,tol", [(0, 3e-7), (1, lp.op_bound(1)), (2, lp.op_bound(2))])      # measured 1.3e-7 / 5.1e-3 / 4.8e-4
def test_fir_resampling_matches_the_reference(golden_dir, dt, tol):
    """upsample_2d / downsample_2d (up_or_down_sampling.py:202-264) on [2,5,16,12]; 16-bit storage: against the golden of the
    rounded input is not available, so the bound is the storage rounding of input and output (2^-8 / 2^-11 relative)."""
    g = np.load(os.path.join(golden_dir, "fir.npz"))
    x = torch.from_numpy(g["x"])
    xd = _nhwc(x, 8, dt)                                             # 16-byte channel vectors: 5 -> 8 channels
    _, up = _fir(xd, dt, None, 0, True)
    _, down = _fir(xd, dt, None, 0, False)
    eu, ed = _rel(_nchw(up, 5), torch.from_numpy(g["up"])), _rel(_nchw(down, 5), torch.from_numpy(g["down"]))
    print(f"fir dtype {dt}: up {eu:.3g} down {ed:.3g}")
    assert eu < tol and ed < tol, (eu, ed)
    assert float(up.float()[..., 5:].abs().max()) == 0.0             # padding channels stay zero


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 12, 16), (1, 64, 40, 128), (3, 48, 34, 32), (2, 24, 20, 8)])
def test_fir_down_strip_walk_equals_the_block_form(dt, shape):
    """The res-block down-sampler in its hot configuration (affine + SiLU, activated and raw output, 16-bit storage) runs as a strip walk
    (fir_down_strip_kernel, round 5: each input activated once per thread, OH % 8 == 0) - its taps accumulate in the block kernel's order, so the two
    forms must agree BIT FOR BIT; the block form is the one pinned against the reference's resampler (test above) and res-block goldens.
    Shapes: borders in both directions, odd output widths, one 8-row strip and several, OH % 8 != 0 (falls back to the block form)."""
    from universal_speech_enhancement_amd.hip_engine import set_option
    B, H, W, Cc = shape
    g = torch.Generator().manual_seed(H * 131 + W)
    x = (torch.randn(B, H, W, Cc, generator=g) * 1.5).to(torch.bfloat16 if dt == 1 else torch.float16).cuda()
    coef = torch.stack([0.5 + torch.rand(B, Cc, generator=g), torch.randn(B, Cc, generator=g) * 0.3], dim=-1).cuda().contiguous()
    outs = {}
    try:
        for mode in (1, 8, 4, 0):                            # by grid size / 8-row strips / 4-row strips / block form
            set_option("fir_strip", mode)
            outs[mode] = _fir(x, dt, coef, 1, False)
    finally:
        set_option("fir_strip", 1)
    for k in (0, 1):
        assert torch.isfinite(outs[0][k].float()).all()
        for mode in (1, 8, 4):
            assert torch.equal(outs[mode][k], outs[0][k]), (shape, dt, k, mode)


@pytest.mark.parametrize("dt", [1, 2])
@pytest.mark.parametrize("shape", [(2, 24, 48, 128, 1), (3, 16, 32, 256, 1), (1, 40, 16, 128, 0), (2, 13, 21, 128, 1), (2, 64, 32, 128, 1)])
def test_pyramid_head_forms_agree_bit_for_bit_and_match_torch(dt, shape):
    """The output-pyramid head (GroupNorm affine + SiLU -> conv3x3 to 4 fp32 channels + the incoming pyramid; ncsnpp.py:437-470) in its two
    schedules: pyr_conv_ws_kernel (round 5: producer / consumer waves walking several tiles per workgroup, the default; walks of many, few
    and single tiles) and pyr_conv_kernel (one tile per workgroup).  Every output accumulates the same products in the same order => BIT-identical fp32 outputs; and all of them against a torch
    fp32 convolution of the activated input rounded to the storage type (bound: accumulation order over K = 9 x C only, 2e-5 of the maximum).
    Shapes: 128 and 256 channels (one / two blocks per tile), ragged map edges, no activation, walks of several tiles per workgroup."""
    from universal_speech_enhancement_amd.hip_engine import set_option
    B, H, W, Cc, act = shape
    tdt = torch.bfloat16 if dt == 1 else torch.float16
    g = torch.Generator().manual_seed(H * 1000 + W + Cc)
    x = (torch.randn(B, H, W, Cc, generator=g) * 1.2).to(tdt)
    coef = torch.stack([0.5 + torch.rand(B, Cc, generator=g), torch.randn(B, Cc, generator=g) * 0.3], dim=-1).contiguous()
    w = torch.randn(4, Cc, 3, 3, generator=g) * (1.0 / (3.0 * Cc ** 0.5))
    bias = torch.randn(4, generator=g) * 0.1
    res = torch.randn(B, H, W, 4, generator=g)
    xd, cd, rd = x.cuda(), coef.cuda(), res.cuda().contiguous()

    def run():
        out = torch.empty(B, H, W, 4, dtype=torch.float32, device="cuda")
        op = UseConvOp()
        op.B, op.H, op.W, op.Cout, op.ntaps, op.act, op.dtype, op.out_dtype, op.variant = B, H, W, 4, 9, act, dt, 0, 0
        op.C0, op.src0, op.C1, op.XC0, op.XC1 = Cc, xd.data_ptr(), 0, 0, 0
        wn = np.ascontiguousarray(w.numpy(), dtype=np.float32); bn = np.ascontiguousarray(bias.numpy(), dtype=np.float32)
        op.w, op.bias, op.coef, op.res = wn.ctypes.data, bn.ctypes.data, cd.data_ptr(), rd.data_ptr()
        op.out_scale, op.out = 1.0, out.data_ptr()
        check(_lib.lib().use_op_conv(C.byref(op), _stream()), "use_op_conv")
        torch.cuda.synchronize()
        return out.cpu()
    outs = {}
    try:
        for name, ws in (("ws", 1), ("ws_few", 7), ("ws_one", 2), ("ws_many", 1000), ("tile", 0)):   # ws_one: one workgroup walks a whole item (rolling halo across strips)
            set_option("pyr_ws", ws)
            outs[name] = run()
    finally:
        set_option("pyr_ws", 1)
    for name in ("ws_few", "ws_one", "ws_many", "tile"):
        assert torch.equal(outs[name], outs["ws"]), (shape, dt, name)
    # torch fp32 reference on the operands as the kernel sees them: weights and the activated input rounded to the storage type
    a = x.float() * coef[:, None, None, :, 0] + coef[:, None, None, :, 1]
    if act:
        a = torch.nn.functional.silu(a)
    a = a.to(tdt).float().permute(0, 3, 1, 2)
    want = torch.nn.functional.conv2d(a, w.to(tdt).float(), bias, padding=1).permute(0, 2, 3, 1) + res
    err = float((outs["ws"] - want).abs().max() / want.abs().max())
    assert err < (2e-3 if dt == 1 else 3e-4), err                    # the device SiLU uses v_exp / v_rcp: the activated value may round to a neighbour


def _run_resblock(g, dt, variant, up=False, down=False, split=None):
    """One ResnetBlockBigGANpp through the HIP operators; split = channels of the first of two concatenated sources."""
    x, temb = torch.from_numpy(g["x"]), torch.from_numpy(g["temb"])
    W = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    B, Cin, H, Wd = x.shape
    Cout = W["Conv_0.weight"].shape[0]
    cop = _pad32(Cout)
    parts = [Cin] if split is None else [split, Cin - split]
    pads = [_pad32(c) for c in parts]
    # input sources (raw) and the GroupNorm_0 affine over the padded concatenation
    offs = np.cumsum([0] + parts)
    srcs = [(_nhwc(x[:, offs[i]:offs[i + 1]], pads[i], dt), pads[i]) for i in range(len(parts))]
    coef_real = _gn_coef(x, W["GroupNorm_0.weight"], W["GroupNorm_0.bias"], Cin)          # [B][Cin][2]
    coef0 = torch.zeros(B, sum(pads), 2, device="cuda")
    po = np.cumsum([0] + pads)
    for i in range(len(parts)):
        coef0[:, po[i]:po[i] + parts[i]] = coef_real[:, offs[i]:offs[i + 1]]

    def padw(w, outp):                                               # [Cout][Cin][..] -> [outp][sum(pads)][..]
        o = torch.zeros(outp, sum(pads), *w.shape[2:])
        for i in range(len(parts)):
            o[:w.shape[0], po[i]:po[i] + parts[i]] = w[:, offs[i]:offs[i + 1]]
        return o.numpy()

    def padv(v, n):
        o = torch.zeros(n); o[:v.shape[0]] = v
        return o

    tv = torch.nn.functional.linear(torch.nn.functional.silu(temb), W["Dense_0.weight"], W["Dense_0.bias"])     # Dense_0(act(temb))
    tvd = torch.zeros(B, cop); tvd[:, :Cout] = tv
    tvd = tvd.cuda().contiguous()
    stats = torch.zeros(B, cop, 2, dtype=torch.int64, device="cuda")
    if up or down:
        assert split is None
        h_act, x_raw = _fir(srcs[0][0], dt, coef0.contiguous(), 1, up)
        H2, W2 = h_act.shape[1:3]
        h1 = _conv(dt, B, H2, W2, [(h_act, pads[0])], None, 0, padw(W["Conv_0.weight"], cop), padv(W["Conv_0.bias"], cop), tvd,
                   None, None, None, 1.0, Cout, stats, variant)
        sc_in = [(x_raw, pads[0])]
    else:
        H2, W2 = H, Wd
        h1 = _conv(dt, B, H, Wd, srcs, coef0.contiguous(), 1, padw(W["Conv_0.weight"], cop), padv(W["Conv_0.bias"], cop), tvd,
                   None, None, None, 1.0, Cout, stats, variant)
        sc_in = srcs
    # GroupNorm_1 from the totals Conv_0 accumulated: groups of the REAL channel count (the padding channels form extra, all-zero groups)
    cpg = Cout // min(Cout // 4, 32)
    coef1 = torch.empty(B, cop, 2, device="cuda")
    g1, b1 = padv(W["GroupNorm_1.weight"], cop).cuda(), padv(W["GroupNorm_1.bias"], cop).cuda()
    check(_lib.lib().use_op_gn_finalize(_ptr(stats), cop, None, 0, _ptr(g1), _ptr(b1), cop // cpg, H2 * W2, 1e-6, _ptr(coef1), B, _stream()),
          "use_op_gn_finalize")
    w1 = torch.zeros(cop, cop, 3, 3); w1[:Cout, :Cout] = W["Conv_1.weight"]
    if "Conv_2.weight" in W:
        bias = padv(W["Conv_1.bias"] + W["Conv_2.bias"], cop)
        w2 = padw(W["Conv_2.weight"][:, :, 0, 0], cop)
        y = _conv(dt, B, H2, W2, [(h1, cop)], coef1, 1, w1.numpy(), bias, None, sc_in, w2, None, 1.0 / SQRT2, Cout, None, variant)
    else:
        y = _conv(dt, B, H2, W2, [(h1, cop)], coef1, 1, w1.numpy(), padv(W["Conv_1.bias"], cop), None, None, None, srcs[0][0], 1.0 / SQRT2,
                  Cout, None, variant)
    if cop > Cout:
        assert float(y.float()[..., Cout:].abs().max()) == 0.0      # padding channels stay zero
    return _nchw(y, Cout), torch.from_numpy(g["y"])


CASES = [("plain", {}), ("widen", {}), ("down", {"down": True}), ("up", {"up": True}), ("cat", {"split": 32})]


@pytest.mark.parametrize("name,kw", CASES)
@pytest.mark.parametrize("dt,tol", [(0, 1e-6), (1, lp.op_bound(1)), (2, lp.op_bound(2))])      # measured <= 4.3e-7 / 5.2e-3 / 5.8e-4
@pytest.mark.parametrize("variant", [0, 1])
def test_resblock_operators_match_the_reference(golden_dir, name, kw, dt, tol, variant):
    """ResnetBlockBigGANpp (layerspp.py:282-314) in its five shapes: plain (residual), widen (1x1 shortcut), down / up (FIR resampling of
    h and x), cat (two concatenated sources).  variant 0 = the kernel the library's dispatcher picks for the shape, 1 = the generic one."""
    g = np.load(os.path.join(golden_dir, f"resblock_{name}.npz"))
    got, want = _run_resblock(g, dt, variant, **kw)
    err = _rel(got, want)
    print(f"resblock_{name} dtype {dt} variant {variant}: {err:.3g}")
    assert err < tol, (name, dt, variant, err)


@pytest.mark.parametrize("dt,tol", [(0, 4e-7), (1, lp.op_bound(1)), (2, lp.op_bound(2))])      # measured 1.6e-7 / 4.8e-3 / 5.5e-4
def test_attention_block_matches_the_reference(golden_dir, dt, tol):
    """AttnBlockpp (layerspp.py:60-93) on [2,32,8,5] (40 tokens, C = 32): GroupNorm in torch, the four NIN as 1x1 convolutions of
    the library (use_op_conv, ntaps 1), softmax(q k^T / sqrt(C)) v by use_op_attention, NIN_3 with the residual and 1/sqrt(2) fused."""
    g = np.load(os.path.join(golden_dir, "attn.npz"))
    x = torch.from_numpy(g["x"])
    W = {k[2:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w.")}
    B, Cc, H, Wd = x.shape
    xd = _nhwc(x, Cc, dt)
    coef = _gn_coef(x, W["GroupNorm_0.weight"], W["GroupNorm_0.bias"], Cc)
    qkv = []
    for i in range(3):                                               # NIN W is [cin][cout]: the conv weight is its transpose
        qkv.append(_conv(dt, B, H, Wd, [(xd, Cc)], coef, 0, W[f"NIN_{i}.W"].t().contiguous().numpy(), W[f"NIN_{i}.b"], None, None, None, None,
                         1.0, Cc, None, 0, ntaps=1))
    N = H * Wd
    hq = torch.empty(B, N, Cc, dtype=TD[dt], device="cuda")
    check(_lib.lib().use_op_attention(_ptr(qkv[0]), _ptr(qkv[1]), _ptr(qkv[2]), _ptr(hq), dt, B, N, Cc, _stream()), "use_op_attention")
    y = _conv(dt, B, H, Wd, [(hq.reshape(B, H, Wd, Cc), Cc)], None, 0, W["NIN_3.W"].t().contiguous().numpy(), W["NIN_3.b"], None, None, None,
              xd, 1.0 / SQRT2, Cc, None, 0, ntaps=1)
    err = _rel(_nchw(y, Cc), torch.from_numpy(g["y"]))
    print(f"attn dtype {dt}: {err:.3g}")
    assert err < tol, err


@pytest.mark.parametrize("dt", [0, 1, 2])
@pytest.mark.parametrize("shape", [(2, 16, 32, 64, 128, 9), (1, 8, 8, 96, 32, 9), (2, 12, 10, 32, 64, 1), (1, 32, 64, 128, 128, 9)])
def test_conv_with_device_weights_equals_conv_with_host_packed_weights(dt, shape):
    """use_op_conv_dev (training path: fp32 parameter tensors in HBM, laid out for the kernels on the device) against use_op_conv (host
    packing) - bit-identical in every storage type; w_mode 1 = the data gradient's operand (flipped, transposed), w_mode 2 = NIN."""
    from universal_speech_enhancement_amd import training as TR
    from universal_speech_enhancement_amd import training_ops as T
    B, H, W, Cin, Cout, ntaps = shape
    g = torch.Generator().manual_seed(Cin * 7 + H)
    x = torch.randn(B, H, W, Cin, generator=g).to(TD[dt]).cuda()
    w = torch.randn(Cout, Cin, *((3, 3) if ntaps == 9 else (1, 1)), generator=g) * 0.1
    b = torch.randn(Cout, generator=g) * 0.1

    def host(xx, ww, bb, nt):
        Bn, Hn, Wn, Ci = xx.shape
        Co = ww.shape[0]
        out = torch.empty(Bn, Hn, Wn, Co, dtype=xx.dtype, device="cuda")
        op = UseConvOp()
        op.B, op.H, op.W, op.C0, op.C1, op.Cout, op.ntaps, op.act, op.dtype, op.out_dtype, op.variant = Bn, Hn, Wn, Ci, 0, Co, nt, 0, dt, dt, 0
        op.src0 = xx.data_ptr()
        wn = np.ascontiguousarray(ww.numpy(), dtype=np.float32); op.w = wn.ctypes.data
        bn = None
        if bb is not None:
            bn = np.ascontiguousarray(bb.numpy(), dtype=np.float32); op.bias = bn.ctypes.data
        op.out_scale, op.out = 1.0, out.data_ptr()
        check(_lib.lib().use_op_conv(C.byref(op), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "use_op_conv")
        return out

    got = TR._conv_dev(x, w.cuda(), b.cuda(), 0, ntaps, Cout)
    torch.cuda.synchronize()
    assert torch.equal(got, host(x, w.reshape(Cout, Cin, *w.shape[2:]) if ntaps == 9 else w.reshape(Cout, Cin), b, ntaps))
    gy = torch.randn(B, H, W, Cout, generator=g).to(TD[dt]).cuda()              # data gradient: w_mode 1 vs the host-built operand
    gx = TR._conv_dev(gy, w.cuda(), None, 1, ntaps, Cin)
    wt = torch.flip(w, (2, 3)).transpose(0, 1).contiguous()
    assert torch.equal(gx, host(gy, wt if ntaps == 9 else wt.reshape(Cin, Cout), None, ntaps))
    if ntaps == 1:                                                               # NIN matrix [Cin][Cout]
        Wn = w.reshape(Cout, Cin).t().contiguous()
        assert torch.equal(TR._conv_dev(x, Wn.cuda(), b.cuda(), 2, 1, Cout), got)
