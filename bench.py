#!/usr/bin/env python3
"""Benchmark of the SGMSE reverse-SDE sampling path on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: the 30-step predictor-corrector sampler
(reverse_diffusion + Langevin x1, snr 0.5 => 60 score-network evaluations) on B=8 synthetic 4 s / 24 kHz utterances
per GPU (BASELINE configs[1]; T=601 frames, padded T'=640, F=512), bf16 storage / fp32 accumulation, whole loop
replayed as one hipGraph, device Philox noise.  Inputs (compressed STFT spectrograms) are resident in HBM before the
timed region; STFT/iSTFT are outside it.  Multi-GPU: utterances are sharded over ranks (weak scaling,
no data-path collective; one weight-blob broadcast at start-up).

Prints ONE JSON line on rank 0.  `roofline` is measured live with HIP events around every launch of the dominant
kernel (the activation-dtype implicit-GEMM conv) during one eager score evaluation on the same stream;
`cpu_baseline` times the CPU oracle (a port of the reference's path, validated against the reference's golden
vectors) on a bounded sample on this box's host cores.
"""
import argparse
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FLOP_PER_FRAME_NFE = 4.1657e9        # 2*MAC per padded frame per score evaluation (F=512)
PEAK_BF16_TFLOPS = 2500.0            # MI355X dense bf16 MFMA (MI355X_MICROARCH.md)
PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_TBPS = 8.0                  # HBM3E (MI355X_MICROARCH.md); ~6.3 TB/s is what a streaming copy reaches


from universal_speech_enhancement_amd.testing.cpu import usable_cores  # noqa: E402


def cpu_baseline(sd_np):
    """CPU oracle timed per SURVEY.md section 8d: BASELINE configs[0] / cfg1 exactly -- ONE 2 s utterance (L=48000, T=301 frames,
    T'=320), N=5 reverse steps, reverse_diffusion + Langevin x1, snr 0.5 => 10 score evaluations, fp32 -- on this box's host
    cores.  The thread count is the best of a short probe (one 64-frame score evaluation per candidate: torch's CPU conv does
    not scale to hundreds of threads on this shape).  Cost per evaluation is constant in t, so the rate is scaled linearly
    in NFE to the metric's 60-NFE sampler."""
    from oracle import ncsnpp_oracle as no
    from oracle import sde_oracle as so
    from universal_speech_enhancement_amd.testing import noise as tn
    avail = usable_cores()
    sd = no.to_torch(sd_np)
    x = torch.from_numpy(tn.complex_normal(1, "cpu_x", (1, 2, 512, 64)))
    best, best_dt = None, 1e30
    with torch.no_grad():
        for th in sorted({min(avail, c) for c in (8, 16, 32, avail)}):
            torch.set_num_threads(th)
            t0 = time.time()
            no.ncsnpp_forward(sd, x, torch.tensor([0.5]))
            d = time.time() - t0
            if d < best_dt:
                best, best_dt = th, d
            if d > 20.0:
                break
        torch.set_num_threads(best)
        n_utt, L, N = 1, 48000, 5
        wav = torch.from_numpy(tn.synth_noisy_speech(n_utt, L, seed=1234))
        draws = [torch.from_numpy(d) for d in tn.sampler_noise(4321, 1 + 2 * N, (n_utt, 1, 512, 320))]
        t0 = time.time()
        _, _, Y, nfe = so.score_model_sample(lambda xx, t: no.ncsnpp_forward(sd, xx, t), wav, N=N, corrector="langevin",
                                             corrector_steps=1, snr=0.5, noise=so.NoiseSource(replay=draws))
        dt = time.time() - t0
        # second leg: the configs[1] SHAPE - 4 s utterances, T' = 640 - at N = 2 (4 NFE); one utterance of the
        # batch of 8 (the batch would take minutes; the CPU path is linear in the batch), scaled linearly in NFE like the first leg
        L2, N2 = 96000, 2
        wav2 = torch.from_numpy(tn.synth_noisy_speech(1, L2, seed=1234))
        draws2 = [torch.from_numpy(d) for d in tn.sampler_noise(4321, 1 + 2 * N2, (1, 1, 512, 640))]
        t0 = time.time()
        _, _, _, nfe2 = so.score_model_sample(lambda xx, t: no.ncsnpp_forward(sd, xx, t), wav2, N=N2, corrector="langevin",
                                              corrector_steps=1, snr=0.5, noise=so.NoiseSource(replay=draws2))
        dt2 = time.time() - t0
    frames = n_utt * (1 + L // 160)
    frame_nfe_per_s = frames * nfe / dt
    frames2 = 1 + L2 // 160
    cfg2 = {"value": round(frames2 * nfe2 / dt2 / 60.0, 4), "unit": "spectrogram-frames/s",
            "sample": f"configs[1] shape: 1 of the 8 utterances x 4 s ({frames2} frames, T'=640), 2-step PC sampler = {nfe2} NFE in {dt2:.1f} s "
                      f"({frames2 * nfe2 / dt2:.1f} frame*NFE/s), scaled linearly in NFE to 60"}
    return {"value": round(frame_nfe_per_s / 60.0, 4), "unit": "spectrogram-frames/s", "cores": best, "kind": "port", "cfg2_shape": cfg2,
            "sample": f"CPU oracle (torch fp32, {best} of {avail} usable cores) on BASELINE configs[0] exactly: 1 utterance x 2 s "
                      f"({frames} frames, T'=320), 5-step PC sampler (reverse_diffusion + langevin x1) = {nfe} NFE in {dt:.1f} s "
                      f"({frame_nfe_per_s:.1f} frame*NFE/s; {frames / dt:.2f} frames/s at this 10-NFE sampler), scaled linearly in "
                      f"NFE to the metric's 60-NFE sampler"}


def pci_bus_id(device_index):
    """hipDeviceGetPCIBusId of a device (what tells two ranks on one GPU from two GPUs), through the HIP runtime torch has loaded."""
    import ctypes
    try:
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device_index)) == 0:
            return buf.value.decode()
    except OSError:
        pass
    p = torch.cuda.get_device_properties(device_index)
    return str(getattr(p, "pci_bus_id", getattr(p, "uuid", f"cuda:{device_index}")))


def train_step_bench(dev, steps=3):
    """One optimisation step (train_step forward + backward + Adam) of NCSN++ Large at the reference's training configuration
    (configs/model/SGMSE_Large.yaml + configs/data/distort.yaml: batch 4, 512 frames x 512 bins, Adam lr 5e-4 weight_decay 1e-7;
    reference model_wrapper.py:147-208) in bf16 mixed precision (16-bit activations and MFMAs incl. the weight gradients, fp32 parameters);
    the same measurement as scripts/train_step_bench.py."""
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    torch.manual_seed(0)
    m = ScoreModel(backbone="ncsnpplarge", sde="ouve", t_eps=3e-2, condition="noisy", n_fft=1022, hop_length=160, num_frames=512,
                   window="hann", sde_input="noisy", precision="fp32").to(dev)
    m.score_net.requires_grad_(True)
    m.score_net.train_precision = "bf16"
    opt = torch.optim.Adam([p for p in m.parameters() if p.requires_grad], lr=5e-4, weight_decay=1e-7)
    Bt, NF = 4, 512
    clean = torch.randn(Bt, (NF - 1) * 160 + 4000, device=dev) * 0.1
    batch = {"clean": clean, "perturbed": clean + 0.05 * torch.randn_like(clean)}

    def step():
        opt.zero_grad(set_to_none=True)
        loss = m.train_step(batch)
        loss.backward()
        opt.step()
        return loss
    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    torch.cuda.synchronize()
    d = (time.perf_counter() - t0) / steps
    assert torch.isfinite(loss.detach()).all()
    return {"ms_per_step": round(d * 1e3, 1), "frames_per_s": round(Bt * NF / d, 1), "batch": Bt, "frames": NF, "precision": "bf16 mixed (fp32 parameters)",
            "steps_timed": steps, "note": "train_step forward + backward + Adam, the reference's training configuration (SGMSE_Large.yaml, batch 4 x 512 frames)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=8, help="utterances per GPU")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--N", type=int, default=30, help="reverse steps")
    ap.add_argument("--corrector", default="langevin")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--subbatch", type=int, default=None, help="sub-batches per score evaluation (library default if unset)")
    ap.add_argument("--stagger-level", type=int, default=None)
    ap.add_argument("--config", type=int, default=None, choices=[1, 3, 4],
                    help="BASELINE.json configs[] preset: 1 = 8 x 4 s, N=30 PC, bf16 (the default, the headline metric); 3 = batch 16, "
                         "N=200, corrector snr 0.5 (long-horizon latency config); 4 = 8 x 4 s per GPU, N=30 PC, fp16 storage (the "
                         "per-GPU workload of the 32-utterance / 4-GPU config)")
    ap.add_argument("--no-power-probe", action="store_true", help="do not sample rocm-smi during the timed steps")
    ap.add_argument("--no-secondary", action="store_true", help="skip the secondary sampler configurations (predictor-only, N=50, fp32)")
    ap.add_argument("--no-secondary-configs", action="store_true", help="skip configs[3] / configs[4] share / training step inside `secondary`")
    ap.add_argument("--opt", action="append", default=[], help="use_set_option name=value (repeatable; same-box A/B of a tuning knob)")
    ap.add_argument("--roofline-only", action="store_true",
                    help="skip the timed sampler steps; run only the per-launch measurement of the dominant kernel (for "
                         "`rocprofv3 --kernel-trace --stats -- python bench.py --roofline-only`, see profiles/README.md)")
    a = ap.parse_args()
    if a.roofline_only:
        a.steps, a.warmup, a.no_cpu_baseline, a.no_secondary = 0, 0, True, True
    if a.config == 3:
        a.batch, a.N = 16, 200
    elif a.config == 4:
        a.precision = "fp16"

    from universal_speech_enhancement_amd import distributed as D
    from universal_speech_enhancement_amd.hip_engine import HipScoreEngine
    from universal_speech_enhancement_amd.sgmse.model_wrapper import ScoreModel
    from universal_speech_enhancement_amd.testing import noise as tn
    from universal_speech_enhancement_amd.testing import weights as tw

    from universal_speech_enhancement_amd.hip_engine import set_option
    if a.subbatch is not None:
        set_option("subbatch", a.subbatch)
    if a.stagger_level is not None:
        set_option("stagger_level", a.stagger_level)
    for kv in a.opt:
        set_option(kv.split("=")[0], int(kv.split("=")[1]))
    rank, world, local = D.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- weights: rank 0 builds + packs, one broadcast of the packed blob (RCCL over xGMI) ----
    sd_np = tw.make_state_dict(1234, **tw.LARGE) if rank == 0 else None
    eng = HipScoreEngine(precision=a.precision, device=local)
    D.broadcast_weights(eng, sd_np, src=0)

    # ---- synthetic 24 kHz noisy speech -> compressed, padded STFT spectrograms resident in HBM ----
    L = int(a.seconds * 24000)
    glue = ScoreModel(backbone="none", condition="noisy", n_fft=1022, hop_length=160, num_frames=512, sde_input="noisy")
    wav = torch.from_numpy(tn.synth_noisy_speech(a.batch, L, seed=1234 + rank * a.batch)).to(dev)
    Y = glue._spectrogram(wav).contiguous()
    B, _, Fq, Tp = Y.shape
    T = 1 + L // 160
    ncorr = 0 if a.corrector == "none" else 1
    nfe = a.N * (1 + ncorr)
    eng.plan(B, Tp)
    eng.set_sampler(a.N, "reverse_diffusion", a.corrector, 1, 0.5, 3e-2, use_graph=not a.no_graph)

    def barrier():
        torch.cuda.synchronize()
        if torch.distributed.is_initialized():                  # (also at world size 1 under a launcher: same calls as the N-GPU run)
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for w in range(a.warmup):
        eng.sample(Y, seed=4321 + w)
    # socket power / shader clock / LIMITER while the timed steps run (rank 0, a sampling thread; nothing on the GPU's queues): the evaluation
    # runs at the package's power limit.  Through amdsmi in-process: power, clock, hotspot temperature and the violation
    # accumulators (PPT = socket power limit, socket / VR / HBM thermal, PROCHOT) - `limit_reasons` = share of the window each was active.
    probe = {"W": [], "MHz": [], "recs": []}
    stop_probe = threading.Event()

    def power_thread():
        try:
            from universal_speech_enhancement_amd.testing import smi
            amdsmi, hnd = smi.smi_open(pci_bus_id(local))      # the device this rank runs on, by PCI address (not amdsmi index 0)
            probe["cap_W"] = smi.power_cap_watts(amdsmi, hnd)
            while not stop_probe.is_set():
                probe["recs"].append(smi.sample(amdsmi, hnd))
                stop_probe.wait(0.25)
            probe["recs"].append(smi.sample(amdsmi, hnd))
            return
        except Exception:                                       # no amdsmi on this box: rocm-smi's text output (power and clock only)
            probe["recs"] = []
        import re
        import shutil
        import subprocess
        if shutil.which("rocm-smi") is None:
            return
        while not stop_probe.is_set():
            try:
                o = subprocess.run(["rocm-smi", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5).stdout
                m = re.search(r"Power \(W\): ([0-9.]+)", o)
                c = re.search(r"sclk.*\((\d+)Mhz\)", o)
                if m and c and float(m.group(1)) > 0:
                    probe["W"].append(float(m.group(1))); probe["MHz"].append(float(c.group(1)))
            except Exception:
                return
            stop_probe.wait(0.5)
    pt = None
    torch.cuda.device_count()                                   # (cached before the sampling thread initialises amdsmi: torch's own count goes through amdsmi too)
    if rank == 0 and not a.no_power_probe and a.steps:
        pt = threading.Thread(target=power_thread, daemon=True); pt.start()
    barrier()
    t0 = time.perf_counter()
    for k in range(a.steps):
        out = eng.sample(Y, seed=4321 + a.warmup + k)
    barrier()
    dt_local = time.perf_counter() - t0
    dt = D.max_over_ranks(dt_local, device=dev)
    # self-verification of an N-GPU line: what RCCL saw (world size after init) and, per rank, its own step time and the device it ran on
    me = {"rank": rank, "local_rank": local, "ms_per_step": round(dt_local / max(a.steps, 1) * 1e3, 2), "device": torch.cuda.get_device_name(local),
          "pci_bus_id": pci_bus_id(local)}
    if torch.distributed.is_initialized():
        ranks_info = [None] * world
        torch.distributed.all_gather_object(ranks_info, me)
        ranks_seen = torch.distributed.get_world_size()
    else:
        ranks_info, ranks_seen = [me], 1
    stop_probe.set()
    if pt is not None:
        pt.join(timeout=6)
    power_probe = None
    if probe["recs"]:
        from universal_speech_enhancement_amd.testing import smi
        sm = smi.summarise(probe["recs"])
        if "socket_W" in sm:
            lr = sm.get("limit_reasons") or {}
            flat = {k: (max(v) if isinstance(v, list) and v else v) for k, v in lr.items()}     # per-XCD arrays -> their maximum
            power_probe = {"socket_W": sm["socket_W"], "sclk_MHz": sm.get("current_gfxclk"), "hotspot_C": sm.get("temperature_hotspot"),
                           "hbm_C": sm.get("temperature_mem"), "samples": sm["samples"], "window_s": sm.get("window_s"), "limit_reasons": flat,
                           "power_cap_W": probe.get("cap_W"),
                           "source": "amdsmi in-process every 0.25 s during the timed steps (samples within 10 % of the maximum power); limit_reasons = "
                                     "share of that window in which the limiter was active (delta of its amdsmi violation accumulator / delta of acc_counter): "
                                     "ppt_pwr = socket power limit, *_thrm = thermal limiters"}
    if power_probe is None and probe["W"]:
        # (the first samples fall into the ramp: keep those within 10 % of the maximum)
        keep = [i for i, w_ in enumerate(probe["W"]) if w_ >= 0.9 * max(probe["W"])]
        power_probe = {"socket_W": round(sum(probe["W"][i] for i in keep) / len(keep), 1), "sclk_MHz": round(sum(probe["MHz"][i] for i in keep) / len(keep)),
                       "samples": len(keep), "limit_reasons": None,
                       "source": "rocm-smi --showpower --showclocks every 0.5 s during the timed steps (samples within 10 % of the maximum)"}
    if a.steps:
        assert torch.isfinite(torch.view_as_real(out)).all(), "non-finite sampler output"
    else:
        out = torch.zeros_like(Y)

    frames = world * B * T * a.steps
    value = frames / dt
    ms_per_step = dt / max(a.steps, 1) * 1e3
    padded_frame_nfe_per_s = world * B * Tp * nfe * a.steps / dt
    tflops_path = padded_frame_nfe_per_s * FLOP_PER_FRAME_NFE / 1e12 / world     # per GPU

    # ---- roofline of the dominant kernel, HIP events per launch on the current stream ----
    x = out
    tvec = torch.full((B,), 0.5, device=dev)
    eng.profile_score(x, Y, tvec)                                   # warm
    conv_ms, conv_flops, conv_bytes, conv_launches, total_ms = eng.profile_score(x, Y, tvec)
    # the HBM-bound kernels of the same evaluation ("glue kernels individually: HBM-bound"): algorithmic bytes
    # (every operand once) / duration between HIP events / 8 TB/s, per kernel class and map, averaged over the launches
    hbm_kernels = []
    groups = {}
    mfma_groups = {}
    by_map = {}
    for name, Hm, Wm, by, ms, fl in eng.profile_aux(with_flops=True):
        if name == "conv_v4":                                       # the dominant kernel's own launches, for the per-map breakdown
            by_map.setdefault((Hm, Wm), []).append((fl, ms))
            continue
        if fl > 0:
            mfma_groups.setdefault((name, Hm, Wm), []).append((fl, ms))
        else:
            groups.setdefault((name, Hm, Wm), []).append((by, ms))
    # the MFMA-bound kernels beside the dominant one, per kernel class and map: algorithmic FLOPs / HIP-event duration / dense peak
    mfma_kernels = []
    for (name, Hm, Wm), v in sorted(mfma_groups.items(), key=lambda kv: -sum(m for _, m in kv[1])):
        fl = sum(f for f, _ in v); ms = sum(m for _, m in v)
        mfma_kernels.append({"kernel": name, "map": f"{Hm}x{Wm}", "launches_per_score": len(v), "algorithmic_gflop_per_launch": round(fl / len(v) / 1e9, 2),
                             "avg_launch_ms": round(ms / len(v), 4), "ms_per_score": round(ms, 3),
                             "achieved_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1)})
    for (name, Hm, Wm), v in sorted(groups.items(), key=lambda kv: (-kv[0][1] * kv[0][2], kv[0][0])):
        if Hm * Wm < 256 * 320:                                     # the smaller maps are hidden behind the other sub-batch's convolutions
            continue
        by = sum(b for b, _ in v) / len(v); ms = sum(m for _, m in v) / len(v)
        # what actually binds the kernel: the fused GroupNorm + SiLU costs two quarter-rate transcendentals per
        # staged element, which puts the FIR down-sampler (each input evaluated 2.25x) and the pyramid heads above their HBM time
        binds = {"fir_down": "valu (fused SiLU: v_exp + v_rcp per element, 1.7x redundant in the strip walk)",
                 "pyr_conv": "tile walk (65 us with SiLU and MFMAs ablated) + valu (fused SiLU, rolling halo 1.125x) on producer waves beside the MFMA waves",
                 "conv_in": "hbm write (tile walk, per-workgroup GroupNorm partial totals)", "fir_up": "hbm"}.get(name, "hbm")
        hbm_kernels.append({"kernel": name, "map": f"{Hm}x{Wm}", "launches_per_score": len(v), "algorithmic_bytes": round(by),
                            "avg_ms": round(ms, 4), "achieved_GBps": round(by / (ms * 1e-3) / 1e9, 1),
                            "frac": round(by / (ms * 1e-3) / (PEAK_HBM_TBPS * 1e12), 4), "binds": binds})
    peak = PEAK_FP32_TFLOPS if a.precision == "fp32" else PEAK_BF16_TFLOPS      # bf16 and fp16 MFMA share the dense peak
    achieved = conv_flops / (conv_ms * 1e-3) / 1e12
    # HBM traffic per launch comes from rocprofv3 PMC passes (FETCH_SIZE x2 on gfx950 + WRITE_SIZE), which cannot be
    # collected from inside the timed process: the committed summary of `scripts/profile.sh pmc <tag>` + `scripts/pmc_to_json.py` on this workload is quoted.
    traffic, traffic_src = None, None
    if a.precision == "bf16" and (B, Tp) == (8, 640):   # the PMC passes were collected on exactly this workload
        import glob
        import hashlib
        pm = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_conv_v[45].json")), key=lambda f: (int(os.path.basename(f).split("_")[0][1:] or 0), f))
        if pm:
            rec = json.load(open(pm[-1]))
            src = os.path.join(ROOT, rec.get("kernel_source", "universal_speech_enhancement_amd/csrc/use_conv_v4.hip"))
            cur = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16] if os.path.exists(src) else None
            if rec.get("kernel_source_sha16") == cur:            # counters of THIS kernel source, else stale: report null
                traffic = round(rec["hbm_bytes_per_launch"])
                traffic_src = os.path.relpath(pm[-1], ROOT)
            else:
                traffic_src = f"stale ({os.path.relpath(pm[-1], ROOT)} was collected on another version of {rec.get('kernel_source')})"
    roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_unit": "HBM bytes per launch",
                "traffic_source": traffic_src,
                "algorithmic_hbm_bytes_per_launch": round(conv_bytes / max(conv_launches, 1)),
                # 16-bit storage: conv_v5_kernel (round 6: conv_v4's tile and pipeline on v_mfma_f32_16x16x32, less energy per FLOP); fp32: conv_v4_kernel
                "kernel": ("use::conv_v5_kernel<%s> (wide-tile implicit-GEMM 3x3 conv of the large maps on 16x16x32 MFMAs, both ACT variants, five epilogue forms)"
                           if a.precision != "fp32" else "use::conv_v4_kernel<%s> (wide-tile implicit-GEMM 3x3 conv of the large maps, both ACT variants)")
                          % ({"bf16": "bf16,bf16,32", "fp16": "f16,f16,32"}.get(a.precision, "f32,f32,16")),
                "measured": "HIP events around every launch of one eager score evaluation with the sub-batches run back to back "
                            "(one launch on the chip at a time; `rocprofv3 --stats -- python bench.py --roofline-only` agrees); in "
                            "the timed region the same launches share the chip with the other sub-batch's kernels",
                "launches_per_score": conv_launches, "avg_launch_ms": round(conv_ms / max(conv_launches, 1), 4),
                "algorithmic_gflop_per_launch": round(conv_flops / max(conv_launches, 1) / 1e9, 2),
                "kernel_time_share_of_eager_score": round(conv_ms / total_ms, 3),   # of one un-pipelined evaluation (sub-batches back to back)
                "whole_path_tflops_per_gpu": round(tflops_path, 1), "whole_path_frac": round(tflops_path / peak, 4),
                "by_map": [{"map": f"{Hm}x{Wm}", "launches_per_score": len(v), "avg_launch_ms": round(sum(m for _, m in v) / len(v), 4),
                            "achieved_TFLOPs": round(sum(f for f, _ in v) / (sum(m for _, m in v) * 1e-3) / 1e12, 1),
                            "frac": round(sum(f for f, _ in v) / (sum(m for _, m in v) * 1e-3) / 1e12 / peak, 4)}
                           for (Hm, Wm), v in sorted(by_map.items(), key=lambda kv: -kv[0][0] * kv[0][1])],
                "hbm_peak_TBps": PEAK_HBM_TBPS, "hbm_kernels": hbm_kernels,
                "kernels": [dict(k, frac=round(k["achieved_TFLOPs"] / peak, 4)) for k in mfma_kernels[:6]]}

    # ---- secondary numbers of SURVEY.md section 8d (outside the timed region above; each its own graph capture + timed steps):
    # the predictor-only sampler at the benchmark's N, the reference's STOCK default (model_wrapper.py:39-40, 262-269: N = 50, corrector
    # "none"), and the fp32 parity mode (the mode the north_star tolerance is stated for) on the same workload
    secondary = None
    cfg_is_1 = (B, a.N, ncorr, a.precision, a.seconds) == (8, 30, 1, "bf16", 4.0)
    if rank == 0 and world == 1 and not a.no_secondary and a.steps:
        def timed(engine, N, corr, steps):
            engine.plan(B, Tp)
            engine.set_sampler(N, "reverse_diffusion", corr, 1, 0.5, 3e-2, use_graph=not a.no_graph)
            engine.sample(Y, seed=1)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for k in range(steps):
                o = engine.sample(Y, seed=2 + k)
            torch.cuda.synchronize()
            d = (time.perf_counter() - t0) / steps
            n = N * (1 if corr == "none" else 2)
            assert torch.isfinite(torch.view_as_real(o)).all()
            return {"N": N, "corrector": corr, "nfe": n, "frames_per_s": round(B * T / d, 1), "ms_per_step": round(d * 1e3, 2),
                    "ms_per_nfe": round(d * 1e3 / n, 3), "steps_timed": steps}
        secondary = {"predictor_only": dict(timed(eng, a.N, "none", 3), note=f"reverse_diffusion alone at the benchmark's N, {a.precision}"),
                     "reference_stock_default": dict(timed(eng, 50, "none", 2), note=f"ScoreModel.sample defaults (N=50, corrector none), {a.precision}")}
        if a.precision != "fp32":
            eng32 = HipScoreEngine(precision="fp32", device=local)
            eng32.load_state_dict(sd_np)
            secondary["fp32_parity_mode"] = dict(timed(eng32, a.N, a.corrector, 1), note="fp32 storage, exact-fp32 MFMA: the mode the parity tolerances are stated for")
            eng32.close()
        if cfg_is_1 and not a.no_secondary_configs:
            # the other single-GPU configurations of BASELINE.json and the training step, each outside the timed region:
            # configs[3] = batch 16, N = 200, Langevin snr 0.5 (400 NFE: one step); configs[4]'s per-GPU share = 8 x 4 s in fp16 storage
            wav16 = torch.from_numpy(tn.synth_noisy_speech(16, L, seed=1234)).to(dev)
            Y16 = glue._spectrogram(wav16).contiguous()

            def timed_cfg(engine, Yc, N, steps):
                Bc = Yc.shape[0]
                engine.plan(Bc, Tp)
                engine.set_sampler(N, "reverse_diffusion", "langevin", 1, 0.5, 3e-2, use_graph=not a.no_graph)
                engine.sample(Yc, seed=1)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for k in range(steps):
                    o = engine.sample(Yc, seed=2 + k)
                torch.cuda.synchronize()
                d = (time.perf_counter() - t0) / steps
                assert torch.isfinite(torch.view_as_real(o)).all()
                return {"batch": Bc, "N": N, "nfe": 2 * N, "frames_per_s": round(Bc * T / d, 1), "ms_per_step": round(d * 1e3, 2),
                        "ms_per_nfe": round(d * 1e3 / (2 * N), 3), "steps_timed": steps}
            secondary["configs3_long_horizon"] = dict(timed_cfg(eng, Y16, 200, 1), note="BASELINE configs[3]: batch 16 x 4 s, N = 200, Langevin snr 0.5, bf16 (`bench.py --config 3`)")
            del Y16, wav16
            eng16 = HipScoreEngine(precision="fp16", device=local)
            eng16.load_state_dict(sd_np)
            secondary["configs4_fp16_share"] = dict(timed_cfg(eng16, Y, a.N, 2), note="BASELINE configs[4] per-GPU share: 8 x 4 s, N = 30 PC, fp16 storage + fp16 MFMA (`bench.py --config 4`)")
            eng16.close()
            try:
                secondary["train_step"] = train_step_bench(dev)
            except Exception as e:                                # the sampler line must not depend on the training path
                secondary["train_step"] = {"error": str(e)[:200]}
        eng.plan(B, Tp)
        eng.set_sampler(a.N, "reverse_diffusion", a.corrector, 1, 0.5, 3e-2, use_graph=not a.no_graph)

    cfg_name = ("configs[1]" if (B, a.N, ncorr, a.precision, a.seconds) == (8, 30, 1, "bf16", 4.0) else
                "configs[3]" if (B, a.N, ncorr, a.precision, a.seconds) == (16, 200, 1, "bf16", 4.0) else
                "configs[4] per-GPU share" if (B, a.N, ncorr, a.precision, a.seconds) == (8, 30, 1, "fp16", 4.0) else "custom")
    if rank == 0:
        res = {
            "metric": "spectrogram-frames/sec through 30-step PC sampler, 24 kHz" if (a.N, ncorr) == (30, 1)
                      else f"spectrogram-frames/sec through {a.N}-step sampler ({nfe} NFE), 24 kHz",
            "value": round(value, 2), "unit": "spectrogram-frames/s", "n_gpus": world, "steps": a.steps,
            "warmup": a.warmup, "ms_per_step": round(ms_per_step, 2), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": a.precision, "data": "synthetic",
            "config": {"workload": f"{cfg_name}: NCSN++ Large score net, batch={B}x{a.seconds:g} s utterances per GPU, "
                                   f"{a.N}-step PC sampler (reverse_diffusion + {a.corrector} x1, snr 0.5, {nfe} NFE), "
                                   f"{a.precision}, hipGraph={'off' if a.no_graph else 'on'}",
                       "global_batch": world * B, "frames_per_utt": T, "padded_frames_per_utt": Tp, "n_freq": Fq,
                       "N": a.N, "nfe": nfe, "parallelism": f"utterance-sharded x{world}"},
            "padded_frame_nfe_per_s": round(padded_frame_nfe_per_s, 1),
            "ranks_seen": ranks_seen, "backend": torch.distributed.get_backend() if torch.distributed.is_initialized() else None,
            "distinct_devices": len({r["pci_bus_id"] for r in ranks_info}),
            "ms_per_step_min_over_ranks": min(r["ms_per_step"] for r in ranks_info), "ms_per_step_max_over_ranks": max(r["ms_per_step"] for r in ranks_info),
            "ranks": ranks_info,
            "roofline": roofline,
        }
        if power_probe is not None:
            res["power_probe"] = power_probe
        if secondary is not None:
            res["secondary"] = secondary
        if world == 1 and not a.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(sd_np)
        print(json.dumps(res), flush=True)
    if torch.distributed.is_initialized():
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
