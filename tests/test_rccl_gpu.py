"""The multi-GPU call path on ONE GPU: process group on the RCCL backend ("nccl") at world size 1, the weight-blob broadcast on the
library-owned device memory, the MAX all-reduce of the timing, one sampler call - and bench.py under torch.distributed.run with
--nproc-per-node 1.  The 8-GPU scaling run is the driver's; this makes sure its first RCCL call is not the first ever."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    return port


_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["USE_ROOT"])
import torch, torch.distributed as dist
from universal_speech_enhancement_amd import distributed as D
from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
from universal_speech_enhancement_amd.testing import weights as tw, noise as tn
rank, world, local = D.init_from_env()
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == "nccl", (rank, world, dist.is_initialized())
sd = tw.make_state_dict(1234, **tw.LARGE)
eng = HipScoreEngine(precision="bf16", device=local)
D.broadcast_weights(eng, sd, src=0)                       # dist.broadcast on a uint8 view of library-owned device memory
blob = eng.weight_blob()
ref = HipScoreEngine(precision="bf16", device=local); ref.load_state_dict(sd)
assert torch.equal(blob, ref.weight_blob()), "the broadcast changed the packed weights"
assert D.max_over_ranks(1.5, device=torch.device("cuda", local)) == 1.5      # all_reduce(MAX) on a float64 CUDA tensor
y = torch.from_numpy(tn.complex_normal(3, "y", (2, 1, 512, 64))).cuda() * 0.5
eng.plan(2, 64); eng.set_sampler(2, "reverse_diffusion", "langevin", 1, 0.5, 3e-2, use_graph=True)
out = eng.sample(y, seed=7)
ref.plan(2, 64); ref.set_sampler(2, "reverse_diffusion", "langevin", 1, 0.5, 3e-2, use_graph=True)
assert torch.isfinite(torch.view_as_real(out)).all() and torch.equal(out, ref.sample(y, seed=7))
dist.barrier(); dist.destroy_process_group()
print("RCCL_WORLD1_OK")
"""


def _env(port):
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), USE_ROOT=ROOT)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return env


def test_rccl_path_at_world_size_one():
    r = subprocess.run([sys.executable, "-c", _WORKER], env=_env(_free_port()), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


def test_bench_under_the_launcher_with_one_process():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "1", "--warmup", "1",
           "--batch", "2", "--seconds", "1", "--N", "2", "--no-cpu-baseline"]
    env = dict(os.environ); env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    res = json.loads(line)
    assert res["n_gpus"] == 1 and res["value"] > 0 and res["config"]["parallelism"] == "utterance-sharded x1"


_TRAIN_WORKER = r"""
import os, sys
sys.path.insert(0, os.environ["USE_ROOT"])
import torch, torch.distributed as dist
from universal_speech_enhancement_amd import distributed as D
from universal_speech_enhancement_amd.sgmse.backbones import BackboneRegistry
from universal_speech_enhancement_amd.testing import noise as tn
rank, world, local = D.init_from_env()
assert dist.get_backend() == "nccl" and world == 1
torch.manual_seed(3)
net = BackboneRegistry.get_by_name("ncsnpp6M")(input_channels=4, precision="fp32", init_scale=1.0).cuda()
net.requires_grad_(True)
x = torch.from_numpy(tn.complex_normal(5, "trn_x", (2, 2, 64, 64))).cuda() * 0.5
lo, hi = D.shard_bounds(x.shape[0], rank, world)                     # each rank's share of the batch
loss = net(x[lo:hi], torch.tensor([0.4, 0.9], device="cuda")[lo:hi]).abs().square().mean()
loss.backward()
before = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
assert not dict(net.named_parameters())["all_modules.0.W"].requires_grad   # the Fourier projection stays frozen (layerspp.py:35)
n = D.allreduce_gradients(net.parameters())                          # the has-gradient mask + 6 M fp32 gradients in one RCCL all-reduce
torch.cuda.synchronize()
assert n == 2 and len(before) > 100
assert dict(net.named_parameters())["all_modules.0.W"].grad is None
assert all(torch.equal(before[k], p.grad) for k, p in net.named_parameters() if k in before)   # mean over one rank = identity
dist.barrier(); dist.destroy_process_group()
print("RCCL_TRAIN_WORLD1_OK")
"""


def test_gradient_allreduce_over_rccl_at_world_size_one():
    """The data-parallel training exchange (distributed.allreduce_gradients) on the RCCL backend: gradients of a taped forward +
    backward, packed into one bucket, all-reduced on the GPU and handed back unchanged at world size 1."""
    r = subprocess.run([sys.executable, "-c", _TRAIN_WORKER], env=_env(_free_port()), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RCCL_TRAIN_WORLD1_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])


# ---- two REAL ranks on the one GPU (RCCL refuses two ranks on one device, gloo does not) -----------------------------------------
_TWO_RANK_WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["USE_ROOT"])
import numpy as np, torch, torch.distributed as dist
from universal_speech_enhancement_amd import distributed as D
from universal_speech_enhancement_amd.data import LoadWavData
from universal_speech_enhancement_amd.SGMSE_module import SGMSEModule
from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
from universal_speech_enhancement_amd.testing import weights as tw
rank, world, local = D.init_from_env(backend="gloo")
assert world == 2 and dist.get_backend() == "gloo"
torch.cuda.set_device(0)
m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=512, window="hann",
               sde_input="noisy", predictor="reverse_diffusion", corrector="langevin", precision="bf16", use_graph=True)
eng = m.score_net.engine(512, "cuda:0")                  # random-init weights of THIS process: must be replaced by rank 0's blob
sd = tw.make_state_dict(1234, **tw.LARGE) if rank == 0 else None      # only rank 0 ever builds the state dict
D.broadcast_weights(eng, sd, src=0)                       # rank 0: pack + upload; rank 1: alloc_weight_blob + receive
crc = int(eng.weight_blob().to(torch.int64).sum().item())
data = LoadWavData(os.environ["USE_IN"], os.environ["USE_OUT"], batch_size=2, rank=rank, world_size=world)
mod = SGMSEModule(Score=m, sampler_kwargs=dict(N=2, corrector_steps=1, snr=0.5, seed=5))
names = []
t0 = __import__("time").perf_counter()
for batch in data.predict_batches("cuda:0"):
    out = mod.predict_step(batch, 0)
    names += list(batch["name"])
torch.cuda.synchronize()
dt = D.max_over_ranks(__import__("time").perf_counter() - t0)
print("TWO_RANK " + json.dumps({"rank": rank, "names": names, "blob_sum": crc, "dt": dt}), flush=True)
dist.barrier(); dist.destroy_process_group()
"""


def test_two_ranks_on_one_gpu_shard_the_files_and_share_rank0s_weights(tmp_path):
    """Replica predict as the reference runs it under DDP (loadwav_datamodule.py:53-60: each rank its share of the files) with TWO real
    processes: rank 0 packs + broadcasts the weight blob, rank 1 receives it into alloc_weight_blob memory from ANOTHER process (gloo,
    host-staged); both walk LoadWavData(rank, world_size) over a 5-file folder through SGMSEModule.predict_step.  Checks: disjoint cover
    of the files, identical blobs, max_over_ranks agrees, and every written file is bit-equal to a single-process run over the same
    batches (the Langevin step couples a LOCAL batch, so the single-process run uses each rank's batches)."""
    from scipy.io import wavfile
    import numpy as np
    import torch
    from universal_speech_enhancement_amd.testing import noise as tn
    src, dst, dst1 = tmp_path / "in", tmp_path / "out2", tmp_path / "out1"
    os.makedirs(src / "sub")
    lens = [9600, 8000, 9000, 7000, 9600]
    wavs = tn.synth_noisy_speech(5, 9600, seed=404)
    for i, L in enumerate(lens):
        wavfile.write(str((src / "sub" if i % 2 else src) / f"utt{i}.wav"), 24000, np.round(wavs[i, :L] * 32767).astype(np.int16))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), USE_ROOT=ROOT,
                   USE_IN=str(src), USE_OUT=str(dst))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, "-c", _TWO_RANK_WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    assert all(p.returncode == 0 for p in procs), [(o[0][-1500:], o[1][-3000:]) for o in outs]
    recs = sorted((json.loads([ln for ln in o[0].splitlines() if ln.startswith("TWO_RANK ")][-1][9:]) for o in outs), key=lambda r: r["rank"])
    assert sorted(recs[0]["names"] + recs[1]["names"]) == [f"utt{i}" for i in range(5)] and not set(recs[0]["names"]) & set(recs[1]["names"])
    assert len(recs[0]["names"]) == 3 and len(recs[1]["names"]) == 2
    assert recs[0]["blob_sum"] == recs[1]["blob_sum"] and recs[0]["dt"] == recs[1]["dt"] > 0
    # single-process run over the same per-rank batches
    from universal_speech_enhancement_amd.data import LoadWavData
    from universal_speech_enhancement_amd.SGMSE_module import SGMSEModule
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    from universal_speech_enhancement_amd.testing import weights as tw
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=512, window="hann",
                   sde_input="noisy", predictor="reverse_diffusion", corrector="langevin", precision="bf16", use_graph=True)
    m.score_net.load_state_dict({k: torch.from_numpy(v) for k, v in tw.make_state_dict(1234, **tw.LARGE).items()})
    mod = SGMSEModule(Score=m, sampler_kwargs=dict(N=2, corrector_steps=1, snr=0.5, seed=5))
    for r in range(2):
        for batch in LoadWavData(str(src), str(dst1), batch_size=2, rank=r, world_size=2).predict_batches("cuda"):
            mod.predict_step(batch, 0)
    n_cmp = 0
    for root, _, files in os.walk(dst1):
        for f in files:
            a = wavfile.read(os.path.join(root, f))[1]
            b = wavfile.read(os.path.join(root, f).replace(str(dst1), str(dst)))[1]
            assert a.shape == b.shape and np.array_equal(a, b), f
            n_cmp += 1
    assert n_cmp == 5
