"""CPU-only checks: the C-ABI library loads and exports every symbol include/use_hip.h declares, the C++
architecture walk agrees with the reference's state-dict layout, error paths report instead of crashing, and the
host-side mirror of the reference interface behaves like the reference's (registries, pad_spec, ScoreModel keys)."""
import ctypes as C
import os
import re
import warnings

import numpy as np
import pytest
import torch

from universal_speech_enhancement_amd import _lib
from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
from universal_speech_enhancement_amd.sgmse.backbones.arch import ncsnpp_param_shapes
from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
from universal_speech_enhancement_amd.sgmse.sampling import CorrectorRegistry, PredictorRegistry
from universal_speech_enhancement_amd.sgmse.sdes import OUVESDE, SDERegistry
from universal_speech_enhancement_amd.sgmse.util.other import pad_spec
from universal_speech_enhancement_amd.sgmse.util.registry import Registry
from universal_speech_enhancement_amd.testing import weights as tw

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _cfg(nf=128, mult=(1, 1, 2, 2, 2, 2, 2), nrb=2, n_freq=512, prec=1):
    cfg = _lib.UseConfig()
    cfg.nf, cfg.n_levels, cfg.num_res_blocks, cfg.n_freq, cfg.precision = nf, len(mult), nrb, n_freq, prec
    for i, m in enumerate(mult):
        cfg.ch_mult[i] = m
    cfg.theta, cfg.sigma_min, cfg.sigma_max = 1.5, 0.05, 0.5
    return cfg


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "use_hip.h")).read()
    declared = set(re.findall(r"\b(use_[a-z_0-9]+)\s*\(", header))
    assert declared, "no prototypes parsed"
    L = C.CDLL(_lib.LIB_PATH)
    for name in declared:
        assert hasattr(L, name), f"{name} declared in use_hip.h but not exported"
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    assert b"gfx950" in _lib.lib().use_version()


def test_architecture_walk_matches_reference_state_dict_layout():
    L = _lib.lib()
    for arch in (tw.LARGE, dict(nf=128, ch_mult=(1, 2, 2, 2), num_res_blocks=1, input_channels=4),
                 dict(nf=64, ch_mult=(1, 2), num_res_blocks=3, input_channels=4)):
        h = C.c_void_p()
        assert L.use_create(C.byref(_cfg(arch["nf"], arch["ch_mult"], arch["num_res_blocks"])), 0, C.byref(h)) == 0
        want = ncsnpp_param_shapes(**arch)
        got = {}
        for i in range(L.use_num_expected_weights(h)):
            name, shape, nd = C.c_char_p(), (C.c_int64 * 4)(), C.c_int()
            assert L.use_expected_weight(h, i, C.byref(name), shape, C.byref(nd)) == 0
            got[name.value.decode()] = tuple(shape[: nd.value])
        assert got == dict(want)
        assert L.use_destroy(h) == 0


def test_error_paths_report_and_do_not_crash():
    L = _lib.lib()
    h = C.c_void_p()
    assert L.use_create(C.byref(_cfg(nf=100)), 0, C.byref(h)) == -1 and b"nf" in L.use_last_error()
    assert L.use_create(C.byref(_cfg(prec=7)), 0, C.byref(h)) == -1
    assert L.use_create(C.byref(_cfg()), 0, C.byref(h)) == 0
    a = np.zeros((3, 3), np.float32)
    shp = (C.c_int64 * 2)(3, 3)
    assert L.use_set_weight(h, b"not.a.weight", a.ctypes.data_as(C.c_void_p), shp, 2) == -1
    assert b"unexpected weight name" in L.use_last_error()
    assert L.use_set_weight(h, b"output_layer.bias", a.ctypes.data_as(C.c_void_p), shp, 2) == -1
    assert L.use_score(h, None, None, None, None, None) == -3            # weights not committed
    assert L.use_num_noise_draws(h) == -3
    with pytest.raises(_lib.UseHipError):
        _lib.check(L.use_plan(h, 0, 64), "use_plan")
    assert L.use_destroy(h) == 0


def test_timesteps_equal_torch_linspace():
    L = _lib.lib()
    for N in (1, 2, 5, 7, 30, 50, 200):
        buf = (C.c_float * N)()
        assert L.use_timesteps(N, 0.03, buf) == 0
        np.testing.assert_array_equal(np.array(buf[:], np.float32), torch.linspace(1, 0.03, N).numpy())


def test_registry_semantics():
    r = Registry("Thing")

    @r.register("a")
    class A:  # noqa: D401
        pass
    assert r.get_by_name("a") is A and r.get_all_names() == ["a"]
    with pytest.raises(ValueError, match="Thing with name 'b' unknown"):
        r.get_by_name("b")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        r.register("a")(A)
        assert any("doubly registered" in str(x.message) for x in w)
    assert {"reverse_diffusion", "euler_maruyama", "none"} <= set(PredictorRegistry.get_all_names())
    assert {"langevin", "ald", "none"} <= set(CorrectorRegistry.get_all_names())
    assert "ouve" in SDERegistry.get_all_names()
    assert {"ncsnpp", "ncsnpplarge"} <= set(BackboneRegistry.get_all_names())


def test_pad_spec_and_stft_glue():
    Y = torch.zeros(2, 1, 512, 61, dtype=torch.complex64)
    assert pad_spec(Y).shape[-1] == 64 and pad_spec(pad_spec(Y)).shape[-1] == 64
    assert pad_spec(torch.zeros(1, 1, 4, 128)).shape[-1] == 128
    m = ScoreModel(backbone="none", condition="noisy", n_fft=1022, hop_length=160, num_frames=512, sde_input="noisy")
    wav = torch.randn(2, 9600)
    S = m.stft(wav)
    assert S.shape == (2, 512, 61)
    back = m.spec_back(m.spec_fwd(S))
    assert torch.allclose(back, S, rtol=1e-4, atol=1e-5)
    assert torch.allclose(m.istft(S, 9600), wav, atol=1e-4)


def test_score_model_state_dict_layout_and_no_cpu_fallback():
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160,
                   num_frames=512, window="hann", sde_input="noisy")
    assert isinstance(m.sde, OUVESDE) and m.predictor == "reverse_diffusion" and m.corrector == "none"
    sd = m.state_dict()
    want = ncsnpp_param_shapes(**tw.LARGE)
    assert [k[len("score_net."):] for k in sd] == list(want)
    assert sum(v.numel() for v in sd.values()) == 64799782
    # CPU tensors must be refused loudly (there is no CPU implementation of the product path)
    with pytest.raises(_lib.UseHipError):
        z = torch.zeros(1, 1, 512, 64, dtype=torch.complex64)
        m(z, torch.ones(1), [z], z)
    # the reference's constructor default condition="both": 6-channel parameter layout, 3 complex inputs (CPU tensors refused)
    both = ScoreModel(backbone="ncsnpp", condition="both")
    assert both.state_dict()["score_net.output_layer.weight"].shape == (2, 6, 1, 1)
    with pytest.raises(ValueError):
        both.score_net(torch.zeros(1, 2, 256, 64, dtype=torch.complex64), torch.ones(1))
    with pytest.raises(_lib.UseHipError):
        both.score_net(torch.zeros(1, 3, 256, 64, dtype=torch.complex64), torch.ones(1))
    with pytest.raises(ValueError):
        ScoreModel(backbone="no_such_backbone", condition="noisy")


def test_ouve_sde_host_math_matches_oracle():
    from oracle import sde_oracle as so
    sde = OUVESDE()
    t = torch.tensor([1.0, 0.5, 0.03])
    np.testing.assert_allclose(sde._std(t).numpy(), so.ouve_std(t).numpy(), rtol=1e-6)
    x, y = torch.randn(3, 1, 4, 4, dtype=torch.complex64), torch.randn(3, 1, 4, 4, dtype=torch.complex64)
    d1, g1 = sde.sde(x, t, y)
    d2, g2 = so.ouve_sde(x, t, y)
    assert torch.equal(d1, d2) and torch.equal(g1, g2)
    sde.N = 30
    f, G = sde.discretize(x, t, y)
    assert torch.allclose(G, g1 * (1 / 30) ** 0.5)


def test_predict_config_composition_and_loader(tmp_path):
    """Hydra-shaped overrides compose like the reference's CLI; the loader reproduces the collate dict."""
    from scipy.io import wavfile
    from universal_speech_enhancement_amd import predict as P
    cfg = P.compose(["model=SGMSE_Large", "ckpt_path=foo.ckpt", "data.data_folder=/in", "model.Score.precision=fp32",
                     "model.sampler_kwargs.N=30", "data.batch_size=2"])
    assert cfg["ckpt_path"] == "foo.ckpt" and cfg["data"]["data_folder"] == "/in" and cfg["data"]["batch_size"] == 2
    assert cfg["model"]["Score"]["precision"] == "fp32" and cfg["model"]["sampler_kwargs"] == {"N": 30}
    assert cfg["model"]["Score"]["n_fft"] == 1022 and cfg["model"]["Score"]["t_eps"] == 3e-2
    src = tmp_path / "in" / "sub"
    src.mkdir(parents=True)
    rng = np.random.RandomState(0)
    wavfile.write(str(src / "a.wav"), 24000, (rng.randn(3000) * 0.1).astype(np.float32))
    wavfile.write(str(tmp_path / "in" / "b.wav"), 48000, (rng.randn(8000, 2) * 3000).astype(np.int16))
    data = P.instantiate({**cfg["data"], "data_folder": str(tmp_path / "in"), "target_folder": str(tmp_path / "out")})
    batches = list(data.predict_batches(device="cpu"))
    assert len(batches) == 1
    b = batches[0]
    assert set(b) == {"perturbed", "name", "sample_length", "sampling_rate", "audio_path", "data_folder", "target_folder"}
    assert sorted(b["name"]) == ["a", "b"] and b["sampling_rate"] == [24000, 24000]
    assert b["perturbed"].shape == (2, 4000) and sorted(b["sample_length"].tolist()) == [3000, 4000]
    assert abs(float(b["perturbed"].abs().max()) - 0.8) < 1e-6
    model = P.instantiate(cfg["model"])
    assert type(model).__name__ == "SGMSEModule" and model.sampler_kwargs == {"N": 30} and model.Score.precision == "fp32"
    # the refine stage: same config keys as the reference's configs/model/LSGAN.yaml, reference key layout
    cfg2 = P.compose(["model=LSGAN", "model.G.precision=fp32"])
    gan = P.instantiate(cfg2["model"])
    assert type(gan).__name__ == "GANModule" and type(gan.G).__name__ == "NCSNPP_Wrapper" and gan.G.target_len == 479 * 160
    keys = list(gan.state_dict())
    assert keys[0] == "G.net.output_layer.weight" and gan.state_dict()["G.net.output_layer.weight"].shape == (2, 2, 1, 1)
    assert "G.net.all_modules.1.weight" in keys and gan.state_dict()["G.net.all_modules.1.weight"].shape == (128, 2, 3, 3)
    assert "G.net.all_modules.2.Dense_0.weight" in keys        # present (and unused) in the unconditional network too
    with pytest.raises(NotImplementedError):
        gan.G({"clean": torch.zeros(1, 10), "perturbed": torch.zeros(1, 10)})


def test_packed_weight_file_is_written_on_the_host(tmp_path):
    """Lightning-checkpoint keys -> packed weight file, no GPU involved; header fields and failure modes."""
    import struct
    from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
    from universal_speech_enhancement_amd.pack_checkpoint import pack
    from universal_speech_enhancement_amd.testing import weights as tw
    sd = {"G.net." + k: torch.from_numpy(v) for k, v in tw.make_state_dict(4321, **tw.REFINE).items()}
    out = str(tmp_path / "refine.usehip")
    pack(sd, out, "LSGAN", "bf16")
    raw = open(out, "rb").read()
    assert raw[:8] == b"USEHIPWB"
    header_bytes, layout, nf, n_levels = struct.unpack_from("<IIii", raw, 8)
    assert nf == 128 and n_levels == 4 and layout >= 3 and len(raw) > header_bytes + 50_000_000
    blob_bytes = struct.unpack_from("<Q", raw, header_bytes - 8)[0]
    assert len(raw) == header_bytes + blob_bytes
    with pytest.raises(KeyError):
        pack({"unrelated": torch.zeros(1)}, out, "LSGAN", "bf16")
    e = HipScoreEngine(precision="bf16", device=0)            # nothing set yet: nothing to save
    with pytest.raises(_lib.UseHipError):
        e.save_weight_blob(str(tmp_path / "empty.usehip"))
    e.close()



def test_train_step_refuses_frame_counts_it_would_have_to_pad():
    """The reference's train_step feeds the UNPADDED num_frames-frame spectrogram to the network (model_wrapper.py:168-171), which only
    closes for multiples of 2^(levels - 1) frames (64 for the 7-level NCSN++, 8 for the 4-level nf = 96 networks); the stand-in must not
    silently zero-pad another count (extra frames in z and in the loss).  The divisor follows the backbone."""
    import torch
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    batch = {"clean": torch.zeros(1, 20000), "perturbed": torch.zeros(1, 20000)}
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", condition="noisy", sde_input="noisy", n_fft=1022, hop_length=160, num_frames=100)
    with pytest.raises(ValueError, match="multiple of 64"):
        m.train_step(dict(batch))
    m = ScoreModel(backbone="ncsnpp6M", sde="ouve", condition="noisy", sde_input="noisy", n_fft=254, hop_length=64, num_frames=100)
    with pytest.raises(ValueError, match="multiple of 8 "):
        m.train_step(dict(batch))
    m = ScoreModel(backbone="ncsnpp6M", sde="ouve", condition="noisy", sde_input="noisy", n_fft=254, hop_length=64, num_frames=104)
    try:                                  # 104 = 13 x 8 frames passes the check (and then needs a GPU: any other error is fine here)
        m.train_step(dict(batch))
    except ValueError as e:
        assert "multiple of" not in str(e)
    except Exception:                     # noqa: BLE001
        pass


def test_training_entry_points_refuse_without_a_tape_or_a_gpu():
    """SGMSEModule.training_step (reference SGMSE_module.py:46-54) needs trainable parameters (they are created frozen); the taped
    network has no CPU implementation and says so; configure_optimizers mirrors the reference's partial-factory contract (:26-40)."""
    import functools
    import torch
    from universal_speech_enhancement_amd.SGMSE_module import SGMSEModule
    from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    m = ScoreModel(backbone="ncsnpp6M", sde="ouve", condition="noisy", sde_input="noisy", n_fft=254, hop_length=64, num_frames=64, precision="fp32")
    mod = SGMSEModule(Score=m, optimizer=functools.partial(torch.optim.SGD, lr=0.1))
    assert not m.score_net.trainable
    with pytest.raises(RuntimeError, match="frozen"):
        mod.training_step({"clean": torch.zeros(1, 6000), "perturbed": torch.zeros(1, 6000)}, 0)
    cfg = mod.configure_optimizers()
    assert list(cfg[0]) == ["optimizer"] and isinstance(cfg[0]["optimizer"], torch.optim.SGD)
    assert sum(p.numel() for g in cfg[0]["optimizer"].param_groups for p in g["params"]) == sum(p.numel() for p in m.parameters())
    with pytest.raises(RuntimeError, match="optimizer"):
        SGMSEModule(Score=m).configure_optimizers()
    net = BackboneRegistry.get_by_name("ncsnpp6M")(input_channels=4, precision="fp32")
    net.requires_grad_(True)
    assert net.trainable
    with pytest.raises(_lib.UseHipError):
        net(torch.zeros(1, 2, 64, 64, dtype=torch.complex64), torch.ones(1))


def test_module_is_a_lightning_module_where_lightning_exists(monkeypatch):
    """The reference's SGMSEModule is a LightningModule (SGMSE_module.py:10): with `lightning` importable the stand-in subclasses it and
    logs what the reference logs (train / val / test loss, lr); without it (this image) it is a plain nn.Module with the same methods."""
    import importlib
    import sys
    import types
    import torch
    import universal_speech_enhancement_amd.SGMSE_module as M
    assert not M.HAS_LIGHTNING and M.SGMSEModule.__mro__[1] is torch.nn.Module       # this image has no Lightning

    class FakeLightningModule(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self._trainer, self.logged = None, []

        def log(self, name, value, **kw):
            self.logged.append((name, float(value), kw))

        def optimizers(self):
            return types.SimpleNamespace(param_groups=[{"lr": 5e-4}])

    fake = types.ModuleType("lightning")
    fake.LightningModule = FakeLightningModule
    monkeypatch.setitem(sys.modules, "lightning", fake)
    try:
        L = importlib.reload(M)
        assert L.HAS_LIGHTNING and issubclass(L.SGMSEModule, FakeLightningModule)
        score = torch.nn.Module()
        score.score_net = types.SimpleNamespace(trainable=True)
        score.train_step = lambda batch: torch.tensor(3.5, requires_grad=True)
        mod = L.SGMSEModule(Score=score)
        assert float(mod.validation_step({}).detach()) == 3.5 and mod.logged == []          # no trainer attached: nothing is logged
        mod._trainer = object()
        loss = mod.training_step({}, 0)
        assert loss.requires_grad
        mod.validation_step({}); mod.test_step({})
        assert [n for n, _, _ in mod.logged] == ["train/loss_Score", "lr", "val/loss_Score", "test/loss_Score"]
        assert mod.logged[0][2] == dict(on_step=True, on_epoch=True, prog_bar=True) and mod.logged[1][1] == 5e-4
    finally:
        monkeypatch.delitem(sys.modules, "lightning")
        importlib.reload(M)
    assert not M.HAS_LIGHTNING


def test_use_hip_opts_environment_is_applied_at_load(monkeypatch):
    """USE_HIP_OPTS="name=value,..." -> use_set_option at load time (A/B runs with an option flipped); unknown names fail loudly."""
    import importlib
    from universal_speech_enhancement_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        pytest.skip("library not built")
    monkeypatch.setenv("USE_HIP_OPTS", "subbatch=3,plan_cache=4")
    monkeypatch.setattr(_lib, "_lib", None)
    assert _lib.lib() is not None
    for bad in ("no_such_option=1", "subbatch", "subbatch=two"):          # unknown name, no value, no integer: every call fails, not only the first
        monkeypatch.setenv("USE_HIP_OPTS", bad)
        monkeypatch.setattr(_lib, "_lib", None)
        for _ in range(2):
            with pytest.raises(_lib.UseHipError):
                _lib.lib()
    monkeypatch.delenv("USE_HIP_OPTS")
    monkeypatch.setattr(_lib, "_lib", None)
    _lib.lib()
