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
