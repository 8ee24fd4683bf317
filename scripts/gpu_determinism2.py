import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import torch
from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
sd = tw.make_state_dict(1234, **tw.LARGE)
m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=512, window="hann",
               sde_input="noisy", predictor="reverse_diffusion", corrector="ald", precision="fp32", use_graph=True)
m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
wav = torch.from_numpy(tn.synth_noisy_speech(3, 9600, seed=9)).cuda()
y1 = wav[:1] * 0.5
z = torch.from_numpy(tn.sampler_noise(6, 5, (1, 1, 512, 64))).cuda()
a = m.sample({"perturbed": y1 / y1.abs().max()}, N=2, corrector_steps=1, snr=0.5, noise=z)["enhanced"].clone()
b = m.sample({"perturbed": y1 / y1.abs().max()}, N=2, corrector_steps=1, snr=0.5, noise=z)["enhanced"].clone()
print("sample twice equal:", torch.equal(a, b), float((a - b).abs().max()))
nf = y1.abs().max().item()
c = m.sample({"perturbed": y1 / nf}, N=2, corrector_steps=1, snr=0.5, noise=z)["enhanced"].clone()
print("tensor-div vs float-div inputs equal:", torch.equal(y1 / y1.abs().max(), y1 / nf), "outputs maxdiff rel", float((a - c).abs().max() / a.abs().max()))
x_hat = m.enhance(y1, predictor="reverse_diffusion", corrector="ald", N=2, corrector_steps=1, snr=0.5, noise=z)
print("enhance vs sample rel:", float((x_hat - a[0].cpu() * nf).abs().max() / (a[0].abs().max() * nf)))
# B=3 then back to B=1 (re-plan) 
Y3 = m._spectrogram(wav)
_ = m.get_pc_sampler("reverse_diffusion", "ald", Y3, N=2, conditioning=[Y3], seed=1)()
d = m.sample({"perturbed": y1 / y1.abs().max()}, N=2, corrector_steps=1, snr=0.5, noise=z)["enhanced"].clone()
print("after re-plan equal:", torch.equal(a, d), float((a - d).abs().max()))
