"""Soak of the training path: 80 Adam steps of NCSN++ large in bf16 mixed precision on one fixed synthetic batch (memory must stay flat, the
loss must fall), then a sampler call on the trained parameters (the sampling engine re-packs them)."""
import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
torch.manual_seed(0)
m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=256, window="hann", sde_input="noisy", precision="bf16").cuda()
m.score_net.requires_grad_(True); m.score_net.train_precision = "bf16"
opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=1e-4)
clean = torch.randn(2, 255 * 160 + 3000, device="cuda") * 0.1
batch = {"clean": clean, "perturbed": clean + 0.05 * torch.randn_like(clean)}
losses = []
for i in range(80):
    opt.zero_grad(set_to_none=True); loss = m.train_step(batch); loss.backward(); opt.step(); losses.append(float(loss.detach()))
    if i in (5, 79): torch.cuda.synchronize(); print("step", i, "alloc GiB", round(torch.cuda.memory_allocated() / 2**30, 3), "reserved", round(torch.cuda.memory_reserved() / 2**30, 3), flush=True)
print("loss first 5 mean", sum(losses[:5]) / 5, "last 5 mean", sum(losses[-5:]) / 5, "finite", all(l == l for l in losses))
# sampling after training: engine re-packs
with torch.no_grad():
    out = m.sample({"perturbed": batch["perturbed"][:, :9600]}, N=2)
print("sample finite", bool(torch.isfinite(out["enhanced"]).all()))
