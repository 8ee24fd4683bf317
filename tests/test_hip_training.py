"""the gradient half of ``SGMSEModule.training_step`` (reference SGMSE_module.py:46-54 -> model_wrapper.py:147-208) through the
differentiable HIP operators (universal_speech_enhancement_amd/training.py), against the gradients the REFERENCE's own loss.backward()
produced for the same clips, t, z, crop start and weights (tests/golden/train_grads_{a,b}.npz, oracle/gen_golden.py: gen_train_grads).
fp32 storage and exact-fp32 MFMA throughout."""
import os

import numpy as np
import pytest
import torch

from universal_speech_enhancement_amd.testing import noise as tnoise
from universal_speech_enhancement_amd.testing import weights as tw

pytestmark = pytest.mark.gpu


def _case(golden_dir, tag):
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    cond, sde_in, arch, lt = {"a": ("noisy", "noisy", "LARGE", "mse"), "b": ("both", "denoised", "LARGE_BOTH", "mae")}[tag]
    g = np.load(os.path.join(golden_dir, f"train_grads_{tag}.npz"))
    gl = dict(np.load(os.path.join(golden_dir, "train_loss.npz")))
    sd = tw.make_state_dict(int(g["weights_seed"]), **getattr(tw, arch))
    assert tw.weights_checksum(sd) == str(g["crc"])
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition=cond, loss_type=lt, n_fft=1022, hop_length=160,
                   num_frames=int(g["num_frames"]), window="hann", sde_input=sde_in, precision="fp32")
    m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m = m.cuda()
    z = torch.from_numpy(tnoise.complex_normal(int(g["z_seed"]), "train_z_" + tag, (2, 1, 512, 64))).cuda()
    batch = {"clean": torch.from_numpy(gl["clean_" + tag]).cuda(), "perturbed": torch.from_numpy(gl["noisy_" + tag]).cuda(),
             "fake": torch.from_numpy(gl["fake_" + tag]).cuda()}
    return m, batch, torch.from_numpy(g["t"]).cuda(), z, int(g["start"]), g


@pytest.mark.parametrize("tag", ["a", "b"])
def test_training_step_gradients_match_the_reference_backward(golden_dir, tag):
    """Every trained parameter of NCSN++ large (616 tensors, 65 M values): gradient norm within 6e-6 of the reference's, every stored
    tensor / corner within 2.5e-5 of its largest entry (about 2x the measured 2.4e-6 / 1.0e-5); the loss itself within 2e-4.  Cases: condition noisy + mse (random excerpt),
    condition both / sde_input denoised + mae (zero padding, 6 input channels)."""
    m, batch, t, z, start, g = _case(golden_dir, tag)
    frozen = m.train_step(batch, t=t, z=z, start=start)                   # parameters are created frozen: forward-only engine value
    assert not frozen.requires_grad
    m.score_net.requires_grad_(True)
    loss = m.train_step(batch, t=t, z=z, start=start)
    assert loss.requires_grad
    want = float(g["loss"])
    assert abs(float(loss.detach()) - want) < 2e-4 * want, (float(loss.detach()), want)
    assert abs(float(frozen) - want) < 2e-4 * want
    loss.backward()
    torch.cuda.synchronize()
    P = dict(m.score_net.named_parameters())
    assert P["all_modules.0.W"].grad is None                             # GaussianFourierProjection.W is not trained (layerspp.py:37)
    worst_n, worst_t, n_checked = 0.0, 0.0, 0
    for key in g.files:
        kind, _, name = key.partition(".")
        if kind not in ("n", "g", "c"):
            continue
        got = P[name].grad
        assert got is not None, name
        if name.endswith("NIN_1.b"):                                       # analytically zero (softmax is invariant to a shift of k): rounding only
            assert float(got.abs().max()) < 1e-4 and float(np.abs(g[key]).max()) < 1e-4
            n_checked += kind == "n"
            continue
        if kind == "n":
            e = abs(float(got.double().norm()) - float(g[key])) / max(float(g[key]), 1e-30)
            worst_n = max(worst_n, e)
            assert e < 6e-6, (name, float(got.double().norm()), float(g[key]))
            n_checked += 1
        else:
            ref = torch.from_numpy(g[key])
            have = got.detach().cpu() if kind == "g" else got.detach()[:4, :4].cpu()
            e = float((have - ref).abs().max()) / max(float(ref.abs().max()), 1e-30)
            worst_t = max(worst_t, e)
            assert e < 2.5e-5, (name, e)
    assert n_checked == sum(1 for p in P.values() if p.grad is not None) and n_checked > 500
    print(f"[measured] train grads {tag}: worst norm rel {worst_n:.3g}, worst tensor rel-to-max {worst_t:.3g}, {n_checked} tensors")


@pytest.mark.parametrize("prec", ["bf16", "fp16"])
def test_mixed_precision_gradients_stay_close_to_the_fp32_reference(golden_dir, prec):
    """score_net.train_precision = "bf16" / "fp16": activations and their gradients in 16 bits, convolutions on the 16-bit MFMA kernels,
    parameters / weight gradients / statistics fp32 - against the reference's fp32 backward (case a).  Bounds: what the REFERENCE's own
    bf16-autocast backward does against its fp32 backward on this very case (tests/lowprec.py, golden/lowprec_reference.npz: loss 2.6e-3,
    gradient norms 1.6e-2, worst per-tensor relative L2 3.5e-2 over 615 tensors), factor 1.25 for loss and norms, 1.5 for the per-tensor
    relative L2 (taken here over the stored tensors and corners only); fp16 = the bf16 figures / 4.  No loss scaling: the gradients of
    this loss sit well inside fp16's range."""
    import lowprec as lp
    tol_loss, tol_norm, tol_tensor = lp.train_bound(prec, "loss_rel", 1.25), lp.train_bound(prec, "norm_rel_max", 1.25), lp.train_bound(prec, "tensor_rell2_max", 1.5)
    m, batch, t, z, start, g = _case(golden_dir, "a")
    m.score_net.requires_grad_(True)
    m.score_net.train_precision = prec
    loss = m.train_step(batch, t=t, z=z, start=start)
    want = float(g["loss"])
    e_loss = abs(float(loss.detach()) - want) / want
    loss.backward()
    torch.cuda.synchronize()
    P = dict(m.score_net.named_parameters())
    worst_n, worst_t, worst_name, tot_num, tot_den = 0.0, 0.0, None, 0.0, 0.0
    for key in g.files:
        kind, _, name = key.partition(".")
        if kind not in ("n", "g", "c") or name.endswith("NIN_1.b"):
            continue
        got = P[name].grad
        assert got is not None and got.dtype == torch.float32 and torch.isfinite(got).all(), name
        if kind == "n":
            worst_n = max(worst_n, abs(float(got.double().norm()) - float(g[key])) / float(g[key]))
        else:
            ref = torch.from_numpy(g[key]).double()
            have = (got.detach().cpu() if kind == "g" else got.detach()[:4, :4].cpu()).double()
            num, den = float((have - ref).square().sum()), float(ref.square().sum())
            if kind == "g" and den > 0:                                  # whole tensors: the per-tensor relative L2 error
                e = (num / den) ** 0.5
                if e > worst_t:
                    worst_t, worst_name = e, name
            tot_num += num; tot_den += den
    print(f"[measured] mixed {prec}: loss rel {e_loss:.3g} (bound {tol_loss:.3g}), worst norm rel {worst_n:.3g} (bound {tol_norm:.3g}), "
          f"worst per-tensor rel-L2 {worst_t:.3g} at {worst_name} (bound {tol_tensor:.3g}), all stored entries rel-L2 {np.sqrt(tot_num / tot_den):.3g}")
    assert e_loss < tol_loss and worst_n < tol_norm and worst_t < tol_tensor


@pytest.mark.parametrize("prec,tol", [("fp32", 1e-4), ("bf16", 3e-2)])
def test_training_forward_matches_the_sampling_engine_and_an_optimiser_step_reaches_it(prec, tol):
    """(nf = 96 network: 96 / 192-channel maps, 24 / 32 GroupNorm groups; fp32 and bf16 mixed precision.)  The taped forward (training.ncsnpp_forward_train) and the sampling engine's fp32 forward are the same network: same output to
    1e-4 on a small configuration; after optimiser steps the loss on the fixed batch goes down and the engine (re-packed from the updated
    parameters) still agrees with the taped forward."""
    from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
    torch.manual_seed(3)
    net = BackboneRegistry.get_by_name("ncsnpp6M")(input_channels=4, precision="fp32", init_scale=1.0).cuda()
    net.train_precision = prec
    x = torch.from_numpy(tnoise.complex_normal(5, "trn_x", (2, 2, 64, 64))).cuda() * 0.5
    target = torch.from_numpy(tnoise.complex_normal(6, "trn_y", (2, 1, 64, 64))).cuda()
    t = torch.tensor([0.4, 0.9], device="cuda")
    with torch.no_grad():
        eng = net(x, t)
    net.requires_grad_(True)
    out = net(x, t)
    assert out.requires_grad and out.shape == eng.shape
    out = out.detach()
    assert float((out - eng).abs().max()) < tol * float(eng.abs().max())
    opt = torch.optim.Adam([p for p in net.parameters() if p.requires_grad], lr=2e-4)
    losses = []
    for _ in range(6):
        opt.zero_grad(set_to_none=True)
        loss = (net(x, t) - target).abs().square().mean()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    assert all(np.isfinite(losses)) and losses[-1] < 0.9 * losses[0], losses
    with torch.no_grad():
        eng2 = net(x, t)                                                   # engine path: weights re-packed after the optimiser steps
    out2 = net(x, t).detach()
    assert float((eng2 - eng).abs().max()) > 1e-3 * float(eng.abs().max())  # the parameters did move
    assert float((out2 - eng2).abs().max()) < tol * float(eng2.abs().max())


def test_module_training_step_and_optimizer_factory():
    """SGMSEModule.training_step returns the loss with its tape; configure_optimizers builds optimiser + scheduler from the
    constructor's partials in Lightning's layout (reference SGMSE_module.py:26-54)."""
    import functools
    from universal_speech_enhancement_amd.SGMSE_module import SGMSEModule
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    torch.manual_seed(1)
    m = ScoreModel(backbone="ncsnpp6M", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=254, hop_length=64, num_frames=64,
                   window="hann", sde_input="noisy", precision="fp32").cuda()
    mod = SGMSEModule(Score=m, optimizer=functools.partial(torch.optim.Adam, lr=1e-4),
                      scheduler=functools.partial(torch.optim.lr_scheduler.ReduceLROnPlateau, factor=0.5, patience=3))
    wav = torch.from_numpy(tnoise.synth_noisy_speech(2, 6000, seed=5)).cuda()
    batch = {"clean": wav, "perturbed": 0.8 * wav + 0.2 * torch.from_numpy(tnoise.synth_noisy_speech(2, 6000, seed=6)).cuda()}
    with pytest.raises(RuntimeError):
        mod.training_step(batch, 0)                                        # frozen parameters: refuse rather than return a tapeless value
    m.score_net.requires_grad_(True)
    cfg = mod.configure_optimizers()[0]
    opt = cfg["optimizer"]
    assert cfg["lr_scheduler"]["monitor"] == "val/loss_Score_epoch" and isinstance(opt, torch.optim.Adam)
    loss = mod.training_step(batch, 0)
    assert loss.requires_grad and torch.isfinite(loss)
    loss.backward()
    opt.step()
    v = mod.validation_step(batch)
    assert not v.requires_grad and torch.isfinite(v)


def test_three_optimiser_steps_track_the_cpu_oracle():
    """The whole loop - taped forward, backward, parameter update, the next forward on the updated parameters - against autograd through
    the CPU oracle with the same initialisation, batch and optimiser (SGD with momentum: linear in the gradients, so that rounding noise in
    near-zero gradients is not amplified the way Adam's normalisation would): losses within 1e-4 at every step, parameters within 1e-3 of
    the largest update after three steps."""
    from oracle import ncsnpp_oracle as no
    from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
    torch.manual_seed(11)
    net = BackboneRegistry.get_by_name("ncsnpp6M")(input_channels=4, precision="fp32", init_scale=1.0).cuda()
    net.requires_grad_(True)
    ref = {k: v.detach().cpu().clone().requires_grad_(k != "all_modules.0.W") for k, v in net.state_dict().items()}
    start = {k: v.detach().clone() for k, v in ref.items()}
    x = torch.from_numpy(tnoise.complex_normal(21, "sgd_x", (2, 2, 64, 64))) * 0.5
    target = torch.from_numpy(tnoise.complex_normal(22, "sgd_y", (2, 1, 64, 64)))
    t = torch.tensor([0.35, 0.8])
    names = [k for k, p in net.named_parameters() if k != "all_modules.0.W"]
    P = dict(net.named_parameters())
    opt_g = torch.optim.SGD([P[k] for k in names], lr=2e-6, momentum=0.9)
    opt_c = torch.optim.SGD([ref[k] for k in names], lr=2e-6, momentum=0.9)
    torch.set_num_threads(max(1, (os.cpu_count() or 2) // 2))
    for step in range(3):
        opt_g.zero_grad(set_to_none=True); opt_c.zero_grad(set_to_none=True)
        lg = (net(x.cuda(), t.cuda()) - target.cuda()).abs().square().mean()
        lc = (no.ncsnpp_forward(ref, x, t, ch_mult=(1, 1, 1, 1), num_res_blocks=1) - target).abs().square().mean()
        assert abs(float(lg.detach()) - float(lc.detach())) < 1e-4 * float(lc.detach()), (step, float(lg.detach()), float(lc.detach()))
        lg.backward(); lc.backward()
        opt_g.step(); opt_c.step()
    moved = max(float((ref[k].detach() - start[k]).abs().max()) for k in names)
    worst = max(float((P[k].detach().cpu() - ref[k].detach()).abs().max()) for k in names)
    print(f"[measured] three SGD steps: largest parameter update {moved:.3g}, largest deviation from the oracle's parameters {worst:.3g}")
    assert moved > 1e-5 and worst < 1e-3 * moved, (moved, worst)
