"""Pins the CPU oracle (oracle/*.py) to the reference: every fixture in tests/golden was produced by
running the reference's own modules (oracle/gen_golden.py, build container only)."""
import os

import numpy as np
import pytest
import torch

from oracle import ncsnpp_oracle as no
from oracle import sde_oracle as so
from universal_speech_enhancement_amd.testing import noise as tnoise
from universal_speech_enhancement_amd.testing import weights as tw
from universal_speech_enhancement_amd.testing.cpu import usable_cores

TOL = dict(rtol=1e-5, atol=1e-5)


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def test_fir_matches_reference(golden_dir):
    g = _load(golden_dir, "fir.npz")
    x = torch.from_numpy(g["x"])
    np.testing.assert_allclose(no.fir_upsample2(x).numpy(), g["up"], **TOL)
    np.testing.assert_allclose(no.fir_downsample2(x).numpy(), g["down"], **TOL)


@pytest.mark.parametrize("case", ["plain", "widen", "down", "up", "cat"])
def test_resblock_matches_reference(golden_dir, case):
    g = _load(golden_dir, f"resblock_{case}.npz")
    sd = {"blk." + k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")}
    y = no.resblock_biggan(torch.from_numpy(g["x"]), torch.from_numpy(g["temb"]), sd, "blk",
                           up=(case == "up"), down=(case == "down"))
    np.testing.assert_allclose(y.numpy(), g["y"], **TOL)


def test_attn_matches_reference(golden_dir):
    g = _load(golden_dir, "attn.npz")
    sd = {"a." + k[2:]: torch.from_numpy(v) for k, v in g.items() if k.startswith("w.")}
    y = no.attn_block(torch.from_numpy(g["x"]), sd, "a")
    np.testing.assert_allclose(y.numpy(), g["y"], **TOL)


@pytest.mark.parametrize("corr", ["none", "langevin", "ald"])
def test_sampler_matches_reference(golden_dir, corr):
    g = _load(golden_dir, f"sampler_rd_{corr}.npz")
    Y, A = torch.from_numpy(g["Y"]), torch.from_numpy(g["A"])
    draws = tnoise.sampler_noise(int(g["noise_seed"]), int(g["n_draws"]), tuple(Y.shape))

    def score(x, t):
        return -(x - 0.8 * Y) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    x, nfe = so.pc_sampler(score, Y, int(g["N"]), "reverse_diffusion", corr, int(g["corrector_steps"]),
                           float(g["snr"]), float(g["eps"]), so.NoiseSource(replay=[torch.from_numpy(d) for d in draws]))
    assert nfe == int(g["nfe"])
    np.testing.assert_allclose(x.numpy(), g["x"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("pred", ["reverse_diffusion", "euler_maruyama"])
def test_probability_flow_sampler_matches_reference(golden_dir, pred):
    """get_pc_sampler(probability_flow=True) in the reference changes NOTHING: Predictor.__init__ stores the flag and builds its reverse
    SDE as sde.reverse(score_fn) (predictors.py:17), so both predictors run the ordinary reverse SDE.  Pinned by outputs of the
    reference's own sampler with the flag set (Langevin corrector x2 in between): the oracle - which has no such flag - reproduces them."""
    g = _load(golden_dir, f"sampler_pf_{pred}.npz")
    Y, A = torch.from_numpy(g["Y"]), torch.from_numpy(g["A"])
    draws = tnoise.sampler_noise(int(g["noise_seed"]), int(g["n_draws"]), tuple(Y.shape))

    def score(x, t):
        return -(x - 0.8 * Y) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    x, nfe = so.pc_sampler(score, Y, int(g["N"]), pred, "langevin", int(g["corrector_steps"]), float(g["snr"]), float(g["eps"]),
                           so.NoiseSource(replay=[torch.from_numpy(d) for d in draws]))
    assert nfe == int(g["nfe"])
    np.testing.assert_allclose(x.numpy(), g["x"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("corr", ["none", "langevin"])
def test_euler_maruyama_sampler_matches_reference(golden_dir, corr):
    """Row a8: EulerMaruyamaPredictor.update_fn + RSDE.sde / rsde_parts, pinned by outputs of the reference's own sampler."""
    g = _load(golden_dir, f"sampler_em_{corr}.npz")
    Y, A = torch.from_numpy(g["Y"]), torch.from_numpy(g["A"])
    draws = tnoise.sampler_noise(int(g["noise_seed"]), int(g["n_draws"]), tuple(Y.shape))

    def score(x, t):
        return -(x - 0.8 * Y) / (0.1 + t[:, None, None, None] ** 2) + 0.05 * A * torch.tanh(x.abs())

    x, nfe = so.pc_sampler(score, Y, int(g["N"]), "euler_maruyama", corr, int(g["corrector_steps"]),
                           float(g["snr"]), float(g["eps"]), so.NoiseSource(replay=[torch.from_numpy(d) for d in draws]))
    assert nfe == int(g["nfe"])
    np.testing.assert_allclose(x.numpy(), g["x"], rtol=1e-5, atol=1e-6)


@pytest.fixture(scope="module")
def large_sd():
    sd = tw.make_state_dict(1234, **tw.LARGE)
    return no.to_torch(sd), tw.weights_checksum(sd)


def test_weight_recipe_is_stable(golden_dir, large_sd):
    g = _load(golden_dir, "forward_large.npz")
    assert str(g["weights_crc"]) == large_sd[1]


def test_forward_large_matches_reference(golden_dir, large_sd):
    g = _load(golden_dir, "forward_large.npz")
    sd = large_sd[0]
    x = torch.from_numpy(g["x"])
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        for tag in ("a", "b"):
            out = no.ncsnpp_forward(sd, x, torch.from_numpy(g["t_" + tag]))
            ref = g["out_" + tag]
            err = np.abs(out.numpy() - ref).max() / np.abs(ref).max()
            assert err < 2e-5, (tag, err)


def test_forward_six_channel_matches_reference(golden_dir):
    """condition="both": NCSNppLarge(input_channels=6) on cat[x, Y, Y_denoised] (model_wrapper.py:43-46, ncsnpp.py:333-347)."""
    g = _load(golden_dir, "both.npz")
    sd_np = tw.make_state_dict(1234, **tw.LARGE_BOTH)
    assert str(g["weights_crc"]) == tw.weights_checksum(sd_np)
    assert sd_np["output_layer.weight"].shape == (2, 6, 1, 1) and sd_np["all_modules.3.weight"].shape == (128, 6, 3, 3)
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        out = no.ncsnpp_forward(no.to_torch(sd_np), torch.from_numpy(g["x"]), torch.from_numpy(g["t"]))
    err = np.abs(out.numpy() - g["fwd"]).max() / np.abs(g["fwd"]).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("tag,cond,sde_in,arch,losses", [("a", "noisy", "noisy", "LARGE", ("mse",)),
                                                          ("b", "both", "denoised", "LARGE_BOTH", ("mse", "mae"))])
def test_train_loss_matches_reference(golden_dir, tag, cond, sde_in, arch, losses):
    """SURVEY 8f4 (forward half): ScoreModel.train_step = the loss validation_step / test_step log, random draws pinned."""
    g = _load(golden_dir, "train_loss.npz")
    sd_np = tw.make_state_dict(int(g["weights_seed"]), **getattr(tw, arch))
    assert tw.weights_checksum(sd_np) == str(g["crc_" + tag])
    sd = no.to_torch(sd_np)
    z = torch.from_numpy(tnoise.complex_normal(int(g["z_seed"]), "train_z_" + tag, (2, 1, 512, 64)))
    torch.set_num_threads(usable_cores())
    for lt in losses:
        with torch.no_grad():
            loss = so.score_model_train_loss(lambda x, t: no.ncsnpp_forward(sd, x, t), torch.from_numpy(g["clean_" + tag]),
                                             torch.from_numpy(g["noisy_" + tag]), torch.from_numpy(g["t_" + tag]), z, int(g["start"]),
                                             fake=torch.from_numpy(g["fake_" + tag]), condition=cond, sde_input=sde_in,
                                             num_frames=int(g["num_frames"]), loss_type=lt)
        assert abs(float(loss) - float(g[f"loss_{tag}_{lt}"])) < 2e-5 * float(g[f"loss_{tag}_{lt}"]), (lt, float(loss))


def test_train_gradients_match_reference_backward(golden_dir):
    """SURVEY 8f4 (gradient half): autograd through the oracle's train-step loss against the reference's own loss.backward()
    (train_grads_a.npz: per-parameter gradient norms, small tensors in full, corners of the large ones) - the fixture the HIP
    training path is held to (tests/test_hip_training.py)."""
    g = _load(golden_dir, "train_grads_a.npz")
    gl = _load(golden_dir, "train_loss.npz")
    sd_np = tw.make_state_dict(int(g["weights_seed"]), **tw.LARGE)
    assert tw.weights_checksum(sd_np) == str(g["crc"])
    sd = {k: (v.requires_grad_(True) if k != "all_modules.0.W" else v) for k, v in no.to_torch(sd_np).items()}
    z = torch.from_numpy(tnoise.complex_normal(int(g["z_seed"]), "train_z_a", (2, 1, 512, 64)))
    torch.set_num_threads(usable_cores())
    loss = so.score_model_train_loss(lambda x, t: no.ncsnpp_forward(sd, x, t), torch.from_numpy(gl["clean_a"]), torch.from_numpy(gl["noisy_a"]),
                                     torch.from_numpy(g["t"]), z, int(g["start"]), condition="noisy", sde_input="noisy",
                                     num_frames=int(g["num_frames"]), loss_type="mse")
    assert abs(float(loss.detach()) - float(g["loss"])) < 2e-5 * float(g["loss"])
    loss.backward()
    n = 0
    for key in g:
        kind, _, name = key.partition(".")
        if kind not in ("n", "g", "c") or name.endswith("NIN_1.b"):          # NIN_1.b: analytically zero gradient (rounding only)
            continue
        got = sd[name].grad
        if kind == "n":
            assert abs(float(got.double().norm()) - float(g[key])) < 2e-5 * float(g[key]), name
            n += 1
        else:
            have = got if kind == "g" else got[:4, :4]
            assert float((have - torch.from_numpy(g[key])).abs().max()) < 1e-4 * float(np.abs(g[key]).max()), name
    assert n > 500


@pytest.mark.slow
def test_sample_e2e_matches_reference(golden_dir, large_sd):
    g = _load(golden_dir, "sample_e2e.npz")
    sd = large_sd[0]
    wav = torch.from_numpy(g["wav"])
    Tp = (1 + wav.shape[1] // 160 + 63) // 64 * 64
    draws = tnoise.sampler_noise(int(g["noise_seed"]), int(g["n_draws"]), (wav.shape[0], 1, 512, Tp))
    with torch.no_grad():
        enh, _, _, nfe = so.score_model_sample(lambda x, t: no.ncsnpp_forward(sd, x, t), wav, N=int(g["N"]),
                                               corrector="langevin", corrector_steps=int(g["corrector_steps"]),
                                               snr=float(g["snr"]),
                                               noise=so.NoiseSource(replay=[torch.from_numpy(d) for d in draws]))
    ref = g["enhanced"]
    err = np.abs(enh.numpy() - ref).max() / np.abs(ref).max()
    assert err < 1e-4, err


def test_refine_generator_matches_reference(golden_dir):
    """LSGAN refine stage: NCSNpp(discriminative=True).forward and the NCSNPP_Wrapper inference branch."""
    g = _load(golden_dir, "refine.npz")
    sd_np = tw.make_state_dict(int(g["weights_seed"]), **tw.REFINE)
    assert str(g["weights_crc"]) == tw.weights_checksum(sd_np)
    sd = no.to_torch(sd_np)
    net = lambda Y: no.ncsnpp_forward(sd, Y, None, ch_mult=tw.REFINE["ch_mult"], num_res_blocks=1, discriminative=True)
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        out = net(torch.from_numpy(g["x"]))
        fake, _, _ = so.refine_generator(net, torch.from_numpy(g["wav"]))
    assert np.abs(out.numpy() - g["out"]).max() / np.abs(g["out"]).max() < 2e-5
    assert np.abs(fake.numpy() - g["fake"]).max() / np.abs(g["fake"]).max() < 1e-4



@pytest.mark.parametrize("name,arch", [("12m", tw.SMALL12M), ("6m", tw.SMALL6M)])
def test_small_variants_match_reference(golden_dir, name, arch):
    """NCSNpp12M / NCSNpp6M (nf = 96, reference ncsnpp.py:527-559): oracle forward vs outputs of the reference's own modules."""
    g = _load(golden_dir, f"forward_{name}.npz")
    sd_np = tw.make_state_dict(int(g["weights_seed"]), **arch)
    assert str(g["weights_crc"]) == tw.weights_checksum(sd_np)
    x = torch.from_numpy(tnoise.complex_normal(int(g["x_seed"]), "small_x", (2, 2, 512, 64))) * 0.5
    torch.set_num_threads(usable_cores())
    with torch.no_grad():
        out = no.ncsnpp_forward(no.to_torch(sd_np), x, torch.from_numpy(g["t"]), ch_mult=arch["ch_mult"], num_res_blocks=arch["num_res_blocks"])
    err = np.abs(out.numpy() - g["out"]).max() / np.abs(g["out"]).max()
    assert err < 2e-5, err


@pytest.mark.parametrize("name", ["plain", "widen", "down", "up"])
def test_oracle_resblock_gradients_match_reference_backward(golden_dir, name):
    """autograd through the oracle's functional res-block reproduces the gradients of the reference's own backward()
    (resblock_grads_*.npz) - the pin of the gradient goldens the HIP backward operators are tested against."""
    f = np.load(os.path.join(golden_dir, f"resblock_{name}.npz"))
    gr = np.load(os.path.join(golden_dir, f"resblock_grads_{name}.npz"))
    sd = {"blk." + k[2:]: torch.from_numpy(f[k]).requires_grad_(True) for k in f.files if k.startswith("w.")}
    x = torch.from_numpy(f["x"]).requires_grad_(True); temb = torch.from_numpy(f["temb"]).requires_grad_(True)
    y = no.resblock_biggan(x, temb, sd, "blk", up=name == "up", down=name == "down")
    (y * torch.from_numpy(gr["gy"])).sum().backward()
    assert float((x.grad - torch.from_numpy(gr["dx"])).abs().max()) < 1e-5 * float(np.abs(gr["dx"]).max())
    assert float((temb.grad - torch.from_numpy(gr["dtemb"])).abs().max()) < 1e-5 * float(np.abs(gr["dtemb"]).max())
    for k, v in sd.items():
        want = gr["d." + k[4:]]
        assert float((v.grad - torch.from_numpy(want)).abs().max()) < 1e-5 * float(np.abs(want).max()), k


def test_oracle_attention_gradients_match_reference_backward(golden_dir):
    f = np.load(os.path.join(golden_dir, "attn.npz"))
    gr = np.load(os.path.join(golden_dir, "attn_grads.npz"))
    sd = {"blk." + k[2:]: torch.from_numpy(f[k]).requires_grad_(True) for k in f.files if k.startswith("w.")}
    x = torch.from_numpy(f["x"]).requires_grad_(True)
    (no.attn_block(x, sd, "blk") * torch.from_numpy(gr["gy"])).sum().backward()
    assert float((x.grad - torch.from_numpy(gr["dx"])).abs().max()) < 1e-5 * float(np.abs(gr["dx"]).max())
    for k, v in sd.items():
        want = gr["d." + k[4:]]
        assert float((v.grad - torch.from_numpy(want)).abs().max()) < 1e-5 * float(np.abs(want).max()), k
