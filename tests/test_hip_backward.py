"""SURVEY 8f4, minimum slice: gradients of one ResnetBlockBigGANpp through the HIP backward operators (use_op_wgrad, use_op_gn_act_bwd,
use_op_colsum, use_op_dense_bwd, and use_op_conv on flipped weights for the data gradients) against the gradients the REFERENCE's own
backward() produced (tests/golden/resblock_grads_{plain,widen,down,up}.npz, oracle/gen_golden.py).  fp32 storage; bound 1e-4 of each tensor's max."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _pad32(c):
    return (c + 31) // 32 * 32


def _nhwc(x, cp):
    B, Cc, H, W = x.shape
    y = torch.zeros(B, H, W, cp)
    y[..., :Cc] = x.permute(0, 2, 3, 1)
    return y.cuda().contiguous()


@pytest.mark.parametrize("name", ["plain", "widen", "down", "up"])
def test_resblock_gradients_match_the_reference_backward(golden_dir, name):
    from universal_speech_enhancement_amd import training_ops as T
    f = np.load(os.path.join(golden_dir, f"resblock_{name}.npz"))
    gr = np.load(os.path.join(golden_dir, f"resblock_grads_{name}.npz"))
    Wt = {k[2:]: torch.from_numpy(f[k]) for k in f.files if k.startswith("w.")}
    x, temb, gy = torch.from_numpy(f["x"]), torch.from_numpy(f["temb"]), torch.from_numpy(gr["gy"])
    B, Cin, H, Wd = x.shape
    Cout = Wt["Conv_0.weight"].shape[0]
    cip, cop = _pad32(Cin), _pad32(Cout)

    def padw(w, o, i):
        z = torch.zeros(o, i, *w.shape[2:]); z[:w.shape[0], :w.shape[1]] = w
        return z

    def padv(v, n):
        z = torch.zeros(n); z[:v.shape[0]] = v
        return z

    W = {"Conv_0.weight": padw(Wt["Conv_0.weight"], cop, cip).numpy(), "Conv_1.weight": padw(Wt["Conv_1.weight"], cop, cop).numpy()}
    if "Conv_2.weight" in Wt:
        W["Conv_2.weight"] = padw(Wt["Conv_2.weight"][:, :, 0, 0], cop, cip).numpy()
    for k, n in (("GroupNorm_0.weight", cip), ("GroupNorm_0.bias", cip), ("GroupNorm_1.weight", cop), ("GroupNorm_1.bias", cop)):
        W[k + ".dev"] = padv(Wt[k], n).cuda()
    Wdense = torch.zeros(cop, 24); Wdense[:Cout] = Wt["Dense_0.weight"]
    W["Dense_0.weight.dev"] = Wdense.cuda().contiguous()
    g0 = cip // (Cin // min(Cin // 4, 32)); g1 = cop // (Cout // min(Cout // 4, 32))       # groups incl. the padding's all-zero groups
    xd, gyd, tembd = _nhwc(x, cip), _nhwc(gy, cop), temb.cuda().contiguous()
    # forward up to Conv_0's output (the activation the backward needs), with the library's forward operator
    a0 = T.gn_act_fwd(xd, W["GroupNorm_0.weight.dev"], W["GroupNorm_0.bias.dev"], g0)
    tv = torch.zeros(B, cop); tv[:, :Cout] = torch.nn.functional.linear(torch.nn.functional.silu(temb), Wt["Dense_0.weight"], Wt["Dense_0.bias"])
    up, down = name == "up", name == "down"
    if up or down:
        a0 = T.fir(a0, up=up)
    h1 = T.conv_fwd(a0, W["Conv_0.weight"], bias=padv(Wt["Conv_0.bias"], cop).numpy(), temb=tv.cuda().contiguous())
    g = T.resblock_backward(xd, h1, tembd, gyd, W, g0, g1, up=up, down=down)
    torch.cuda.synchronize()

    def rel(got, want):
        return float((got - want).abs().max() / want.abs().max())

    checks = {"x": (g["x"].cpu()[..., :Cin].permute(0, 3, 1, 2), gr["dx"]), "temb": (g["temb"].cpu(), gr["dtemb"]),
              "Conv_0.weight": (g["Conv_0.weight"].cpu()[:Cout, :Cin], gr["d.Conv_0.weight"]), "Conv_0.bias": (g["Conv_0.bias"].cpu()[:Cout], gr["d.Conv_0.bias"]),
              "Conv_1.weight": (g["Conv_1.weight"].cpu()[:Cout, :Cout], gr["d.Conv_1.weight"]), "Conv_1.bias": (g["Conv_1.bias"].cpu()[:Cout], gr["d.Conv_1.bias"]),
              "GroupNorm_0.weight": (g["GroupNorm_0.weight"].cpu()[:Cin], gr["d.GroupNorm_0.weight"]),
              "GroupNorm_0.bias": (g["GroupNorm_0.bias"].cpu()[:Cin], gr["d.GroupNorm_0.bias"]),
              "GroupNorm_1.weight": (g["GroupNorm_1.weight"].cpu()[:Cout], gr["d.GroupNorm_1.weight"]),
              "GroupNorm_1.bias": (g["GroupNorm_1.bias"].cpu()[:Cout], gr["d.GroupNorm_1.bias"]),
              "Dense_0.weight": (g["Dense_0.weight"].cpu()[:Cout], gr["d.Dense_0.weight"]), "Dense_0.bias": (g["Dense_0.bias"].cpu()[:Cout], gr["d.Dense_0.bias"])}
    if "Conv_2.weight" in W:
        checks["Conv_2.weight"] = (g["Conv_2.weight"].cpu()[:Cout, :Cin], gr["d.Conv_2.weight"][:, :, 0, 0])
        checks["Conv_2.bias"] = (g["Conv_2.bias"].cpu()[:Cout], gr["d.Conv_2.bias"])
    worst = 0.0
    for k, (got, want) in checks.items():
        e = rel(got, torch.from_numpy(np.asarray(want)))
        print(f"[measured] resblock_{name} grad {k}: {e:.3g}")
        worst = max(worst, e)
        assert e < 1e-4, (name, k, e)


def test_attention_block_gradients_match_the_reference_backward(golden_dir):
    """AttnBlockpp (layerspp.py:60-93) on [2,32,8,5]: gradients through use_op_attention_bwd (core), use_op_wgrad / use_op_conv (the four
    NIN as 1x1 convolutions) and use_op_gn_act_bwd (GroupNorm without activation) against the reference's own backward()."""
    from universal_speech_enhancement_amd import training_ops as T
    f = np.load(os.path.join(golden_dir, "attn.npz"))
    gr = np.load(os.path.join(golden_dir, "attn_grads.npz"))
    x, gy = torch.from_numpy(f["x"]), torch.from_numpy(gr["gy"])
    Cc = x.shape[1]
    W = {k[2:]: f[k] for k in f.files if k.startswith("w.")}
    for k in ("GroupNorm_0.weight", "GroupNorm_0.bias"):
        W[k + ".dev"] = torch.from_numpy(W[k]).cuda()
    g = T.attn_block_backward(_nhwc(x, Cc), _nhwc(gy, Cc), W, min(Cc // 4, 32))
    torch.cuda.synchronize()
    checks = {"x": (g["x"].cpu().permute(0, 3, 1, 2), gr["dx"]), "GroupNorm_0.weight": (g["GroupNorm_0.weight"].cpu(), gr["d.GroupNorm_0.weight"]),
              "GroupNorm_0.bias": (g["GroupNorm_0.bias"].cpu(), gr["d.GroupNorm_0.bias"])}
    for i in range(4):
        checks[f"NIN_{i}.W"] = (g[f"NIN_{i}.W"].cpu(), gr[f"d.NIN_{i}.W"]); checks[f"NIN_{i}.b"] = (g[f"NIN_{i}.b"].cpu(), gr[f"d.NIN_{i}.b"])
    for k, (got, want) in checks.items():
        want = torch.from_numpy(np.asarray(want))
        if float(want.abs().max()) < 1e-6:                   # NIN_1.b: a key bias shifts every score of a row alike - its gradient is exactly 0
            print(f"[measured] attn grad {k}: |got| {float(got.abs().max()):.3g}, |reference| {float(want.abs().max()):.3g} (analytically zero)")
            assert float(got.abs().max()) < 1e-5, k
            continue
        e = float((got - want).abs().max() / want.abs().max())
        print(f"[measured] attn grad {k}: {e:.3g}")
        assert e < 1e-4, (k, e)


@pytest.mark.parametrize("dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,H,W,Cout,Cin,ntaps", [(2, 12, 10, 32, 32, 9), (1, 16, 8, 64, 96, 9), (3, 4, 1, 128, 64, 9), (2, 32, 48, 96, 128, 9),
                                                  (2, 24, 6, 64, 64, 1), (1, 64, 64, 128, 128, 9), (2, 8, 3, 32, 64, 9), (2, 4, 2, 256, 128, 1), (1, 2, 1, 64, 64, 1)])
def test_weight_gradient_on_the_16bit_matrix_pipe(dt, B, H, W, Cout, Cin, ntaps):
    """wgrad16_kernel (16-bit MFMA, transposing LDS reads) against (a) the fp32-MFMA kernel on the same 16-bit tensors (identical
    products, fp32 accumulation: agreement to summation order, 2e-6 of the largest entry) and (b) torch's fp64 contraction of them."""
    from universal_speech_enhancement_amd import training_ops as T
    from universal_speech_enhancement_amd.hip_engine import set_option
    g = torch.Generator().manual_seed(H * 131 + W)
    dy = (torch.randn(B, H, W, Cout, generator=g) * 0.5).to(dt).cuda()
    x = torch.randn(B, H, W, Cin, generator=g).to(dt).cuda()
    try:
        set_option("wgrad_mfma16", 1)
        dw16, db16 = T.conv_wgrad(dy, x, ntaps=ntaps, alpha=0.75)
        set_option("wgrad_mfma16", 0)
        dw32, db32 = T.conv_wgrad(dy, x, ntaps=ntaps, alpha=0.75)
    finally:
        set_option("wgrad_mfma16", 1)
    torch.cuda.synchronize()
    dyd, xd = dy.double().cpu(), x.double().cpu()
    if ntaps == 9:
        xp = torch.nn.functional.pad(xd, (0, 0, 1, 1, 1, 1))
        ref = torch.stack([torch.einsum("bhwo,bhwi->oi", dyd, xp[:, ky:ky + H, kx:kx + W]) for ky in range(3) for kx in range(3)], dim=-1)
        ref = ref.view(Cout, Cin, 3, 3) * 0.75
    else:
        ref = torch.einsum("bhwo,bhwi->oi", dyd, xd) * 0.75
    scale = float(ref.abs().max())
    assert torch.isfinite(dw16).all() and torch.isfinite(db16).all()
    assert float((dw16 - dw32).abs().max()) < 2e-6 * scale
    assert float((dw16.double().cpu() - ref).abs().max()) < 2e-6 * scale
    refb = dyd.sum((0, 1, 2)) * 0.75
    assert float((db16.double().cpu() - refb).abs().max()) < 2e-6 * float(refb.abs().max()) + 1e-6
    assert float((db16 - db32).abs().max()) < 2e-6 * float(refb.abs().max()) + 1e-6


@pytest.mark.parametrize("dt,tol", [(torch.float32, 2e-6), (torch.bfloat16, 1.2e-2), (torch.float16, 1.5e-3)])
@pytest.mark.parametrize("B,H,W,Cc,G,act", [(2, 12, 10, 32, 8, 1), (1, 64, 64, 128, 32, 1), (2, 4, 1, 256, 32, 0), (1, 8, 8, 96, 24, 1), (3, 16, 32, 384, 32, 1)])
def test_groupnorm_silu_operators_in_every_storage_type(dt, tol, B, H, W, Cc, G, act):
    """use_op_gn_act_fwd / use_op_gn_act_bwd (sliced reductions, statistics reused from the forward workspace, fused `add`) against
    torch's autograd on the same (already rounded) tensors in fp64; 16-bit bounds = one storage rounding of the result."""
    from universal_speech_enhancement_amd import training_ops as T
    g = torch.Generator().manual_seed(B * 100 + Cc)
    x = (torch.randn(B, H, W, Cc, generator=g) * 1.5 + 0.3).to(dt)
    dy = torch.randn(B, H, W, Cc, generator=g).to(dt)
    add = torch.randn(B, H, W, Cc, generator=g).to(dt)
    gamma, beta = torch.randn(Cc, generator=g) * 0.5 + 1.0, torch.randn(Cc, generator=g) * 0.2
    xr = x.double().permute(0, 3, 1, 2).requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    yr = torch.nn.functional.group_norm(xr, G, gr, br, eps=1e-6)
    if act:
        yr = torch.nn.functional.silu(yr)
    yr.backward(dy.double().permute(0, 3, 1, 2))
    y, work = T.gn_act_fwd(x.cuda(), gamma.cuda(), beta.cuda(), G, act=act, return_work=True)
    dx, dg, db = T.gn_act_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), G, act=act, add=add.cuda(), add_scale=0.5, fwd_work=work)
    dx2, dg2, db2 = T.gn_act_bwd(x.cuda(), dy.cuda(), gamma.cuda(), beta.cuda(), G, act=act)      # statistics recomputed, no add
    torch.cuda.synchronize()

    def rel(a, b_):
        return float((a.double().cpu() - b_).abs().max()) / float(b_.abs().max())
    want_dx = xr.grad.permute(0, 2, 3, 1)
    assert rel(y, yr.detach().permute(0, 2, 3, 1)) < tol
    assert rel(dx2, want_dx) < tol
    assert rel(dx, want_dx + 0.5 * add.double()) < tol
    assert rel(dg, gr.grad) < max(tol * 0.1, 2e-6) and rel(db, br.grad) < max(tol * 0.1, 2e-6)     # parameter gradients: fp32 sums of fp32 terms
    assert torch.equal(dg, dg2) and torch.equal(db, db2)
