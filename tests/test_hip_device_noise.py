"""GPU pin of the branch bench.py times: the fused sampler drawing its noise ON THE DEVICE (use_sample(noise = NULL, seed)).

The reference draws torch.randn_like in the order prior (sdes.py:248-254) -> per reverse step the corrector draw(s)
(sampling/correctors.py:54) -> the predictor draw (sampling/predictors.py:63).  Every other oracle comparison in tests/ injects
recorded noise; here the device stream itself is pinned:

  (a) use_fill_noise(seed, d) exposes draw d of that stream, and use_sample(noise = [fill(seed, d) for d]) must reproduce
      use_sample(noise = NULL, seed) BIT FOR BIT - fp32 and bf16, hipGraph replay and eager launches, Langevin (whose norms kernel
      and update kernel must see the SAME z) and ALD, on B = 8 evaluated as 3 + 3 + 2 items on three streams;
  (b) the 61 draws of the benchmark configuration (N = 30, Langevin x 1) are fresh streams: every pair of draws and every pair of batch
      items is uncorrelated, and every draw has the moments of a standard complex normal;
  (c) the replayed run then matches the CPU oracle fed the same recorded draws (fp32, <= 2e-3 per item), so the device-noise run IS
      the reference's sampler on a known noise sequence.
"""
import numpy as np
import pytest
import torch

from oracle import ncsnpp_oracle as no
from oracle import sde_oracle as so
from universal_speech_enhancement_amd.testing import noise as tnoise
from universal_speech_enhancement_amd.testing import weights as tw
from universal_speech_enhancement_amd.testing.cpu import usable_cores

pytestmark = pytest.mark.gpu

B, TP, N_STEPS, SEED = 8, 64, 3, 20260929
GAINS = (1.0, 5.0, 0.2, 2.5, 0.5, 1.5, 0.1, 3.0)


@pytest.fixture(scope="module")
def sd_np():
    return tw.make_state_dict(1234, **tw.LARGE)


@pytest.fixture(scope="module")
def engines(sd_np):
    from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
    made = {}

    def get(prec):
        if prec not in made:
            e = HipScoreEngine(precision=prec)
            e.load_state_dict(sd_np)
            e.plan(B, TP)
            made[prec] = e
        return made[prec]
    yield get
    for e in made.values():
        e.close()


def _Y():
    y = torch.from_numpy(tnoise.complex_normal(77, "Y", (B, 1, 512, TP))) * torch.tensor(GAINS).view(B, 1, 1, 1) * 0.3
    return y.cuda()


@pytest.mark.parametrize("prec", ["fp32", "bf16"])
@pytest.mark.parametrize("corrector", ["langevin", "ald"])
def test_device_noise_run_equals_the_replay_of_its_own_draws(engines, prec, corrector):
    """(a): bit equality of the two noise branches of use_sample, graph on and off."""
    eng = engines(prec)
    Y = _Y()
    outs = {}
    for use_graph in (True, False):
        eng.set_sampler(N_STEPS, "reverse_diffusion", corrector, 1, 0.5, 3e-2, use_graph=use_graph)
        nd = eng.num_noise_draws()
        assert nd == 1 + 2 * N_STEPS
        dev = eng.sample(Y, seed=SEED)
        draws = torch.stack([eng.fill_noise(SEED, d, Y.shape) for d in range(nd)])
        rep = eng.sample(Y, noise=draws)
        assert torch.isfinite(torch.view_as_real(dev)).all()
        assert torch.equal(dev, rep), f"{prec} {corrector} graph={use_graph}: device-noise run != replay of use_fill_noise draws"
        # a second device-noise call on the same (replayed) graph draws the same stream again; another seed another one
        assert torch.equal(eng.sample(Y, seed=SEED), dev)
        assert not torch.equal(eng.sample(Y, seed=SEED + 1), dev)
        # negative control: the replay with two draws swapped (corrector <-> predictor of step 0) must differ - the order of consumption is pinned
        sw = draws.clone(); sw[1], sw[2] = draws[2], draws[1]
        assert not torch.equal(eng.sample(Y, noise=sw), dev)
        outs[use_graph] = dev
    assert torch.equal(outs[True], outs[False]), "hipGraph replay must be bit-identical to eager launches"


def test_benchmark_configuration_draws_are_fresh_independent_standard_normals(engines):
    """(b): N = 30, Langevin x 1 => 61 draws of [8, 1, 512, 64].  Bounds in units of the sampling error of the statistic."""
    eng = engines("fp32")
    eng.set_sampler(30, "reverse_diffusion", "langevin", 1, 0.5, 3e-2, use_graph=False)
    nd = eng.num_noise_draws()
    assert nd == 61
    shape = (B, 1, 512, TP)
    D = torch.stack([torch.view_as_real(eng.fill_noise(SEED, d, shape)).reshape(-1) for d in range(nd)]).double()   # [61, n] reals
    n = D.shape[1]
    # moments of every draw (complex normal with E|z|^2 = 1: re, im ~ N(0, 1/2), independent)
    mean = D.mean(1); var = D.var(1)
    re, im = D[:, 0::2], D[:, 1::2]
    assert float(mean.abs().max()) < 5e-3 and float((var - 0.5).abs().max()) < 5e-3
    assert float((re * im).mean(1).abs().max()) < 5e-3
    kurt = ((D - mean[:, None]) ** 4).mean(1) / var ** 2
    assert float((kurt - 3.0).abs().max()) < 0.1
    # every pair of draws: |correlation| < 5 / sqrt(n) (n = 524 288 reals: 6.9e-3; 1 830 pairs, expected maximum ~ 3.6 / sqrt(n) = 5e-3)
    Z = (D - mean[:, None]) / D.std(1)[:, None]
    Cd = (Z @ Z.T) / n
    off = Cd - torch.diag(torch.diag(Cd))
    bound_d = 5.0 / np.sqrt(n)
    print(f"[measured] draw-pair correlations: max |r| {float(off.abs().max()):.2e} (bound {bound_d:.2e}); mean {float(mean.abs().max()):.1e} var-0.5 {float((var - 0.5).abs().max()):.1e}")
    assert float(off.abs().max()) < bound_d
    # draw d must not be draw d' shifted by an element or by an item (a counter mix-up): lag-1 and lag-one-item correlations
    for lag in (2, n // B):
        r = (Z[:, :-lag] * Z[:, lag:]).mean(1)
        assert float(r.abs().max()) < bound_d, (lag, float(r.abs().max()))
    # every pair of batch items (and with them the sub-batches 0-2 / 3-5 / 6-7), over all draws: 61 * 65 536 reals per item
    per = D.reshape(nd, B, -1).permute(1, 0, 2).reshape(B, -1)
    Zi = (per - per.mean(1, keepdim=True)) / per.std(1, keepdim=True)
    Ci = (Zi @ Zi.T) / Zi.shape[1]
    offi = Ci - torch.diag(torch.diag(Ci))
    bound_i = 5.0 / np.sqrt(Zi.shape[1])
    print(f"[measured] item-pair correlations: max |r| {float(offi.abs().max()):.2e} (bound {bound_i:.2e} < 5e-3)")
    assert bound_i < 5e-3 and float(offi.abs().max()) < bound_i
    # the same draw index under another seed is another stream
    other = torch.view_as_real(eng.fill_noise(SEED + 1, 0, shape)).reshape(-1).double()
    assert abs(float(((other - other.mean()) / other.std() * Z[0]).mean())) < bound_d


def test_device_noise_run_matches_the_oracle_on_its_recorded_draws(sd_np):
    """(c): ScoreModel.sample(seed = s) == ScoreModel.sample(noise = recorded draws) bit for bit, and the oracle's ScoreModel.sample on the
    CPU with those draws agrees per item to 2e-3 (fp32; the bound of the injected-noise tests in test_hip_fused_batch.py)."""
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=512,
                   window="hann", sde_input="noisy", predictor="reverse_diffusion", corrector="langevin", precision="fp32", use_graph=True)
    m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd_np.items()})
    wav = torch.from_numpy(tnoise.synth_noisy_speech(B, 9600, seed=321)) * torch.tensor(GAINS).view(B, 1)
    dev = m.sample({"perturbed": wav.cuda()}, N=N_STEPS, corrector_steps=1, snr=0.5, seed=SEED)["enhanced"].cpu()
    eng = m.score_net.engine(512, torch.device("cuda"), sde_constants=(m.sde.theta, m.sde.sigma_min, m.sde.sigma_max))
    draws = torch.stack([eng.fill_noise(SEED, d, (B, 1, 512, TP)) for d in range(1 + 2 * N_STEPS)])
    rep = m.sample({"perturbed": wav.cuda()}, N=N_STEPS, corrector_steps=1, snr=0.5, noise=draws)["enhanced"].cpu()
    assert torch.equal(dev, rep)
    torch.set_num_threads(usable_cores())
    sd = no.to_torch(sd_np)
    with torch.no_grad():
        ref, _, _, nfe = so.score_model_sample(lambda xx, t: no.ncsnpp_forward(sd, xx, t), wav, N=N_STEPS, predictor="reverse_diffusion",
                                               corrector="langevin", corrector_steps=1, snr=0.5,
                                               noise=so.NoiseSource(replay=[d.cpu() for d in draws]))
    assert nfe == 2 * N_STEPS
    for b in range(B):
        err = float((dev[b] - ref[b]).abs().max() / ref[b].abs().max())
        print(f"[measured] device-noise sampler vs oracle on the recorded draws, item {b} (gain {GAINS[b]}): {err:.3g} (bound 2e-3)")
        assert err < 2e-3, (b, err)
