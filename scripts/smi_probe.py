#!/usr/bin/env python3
"""What limits the shader clock while a workload runs: samples the amdsmi violation accumulators (PPT / thermal / VR / HBM / PROCHOT,
per-XCP "gfx clock below host limit" by power and by temperature), gpu_metrics throttle bits, socket power, hotspot temperature and
clocks around and during a command.

    python scripts/smi_probe.py [--dump] [--period 0.5] -- <command ...>

--dump prints one full record of every query first (bring-up: which fields this driver fills in).  Prints one JSON summary line at
the end: per limiter the share of the sampling window in which it was active (delta of its accumulator / delta of acc_counter).
Used by scripts/profile.sh power and by bench.py's power_probe (the same queries, in-process)."""
import json
import subprocess
import sys
import time


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from universal_speech_enhancement_amd.testing.smi import smi_open, sample, summarise, _try  # noqa: E402


def main():
    args = sys.argv[1:]
    dump = "--dump" in args
    period = 0.5
    if "--period" in args:
        period = float(args[args.index("--period") + 1])
    cmd = args[args.index("--") + 1:] if "--" in args else None
    amdsmi, h = smi_open()
    if dump:
        for name in ("amdsmi_get_violation_status", "amdsmi_get_power_info", "amdsmi_get_power_cap_info", "amdsmi_get_gpu_metrics_info",
                     "amdsmi_get_clock_info"):
            f = getattr(amdsmi, name, None)
            if f is None:
                print(name, "absent"); continue
            a = (h, amdsmi.AmdSmiClkType.GFX) if name == "amdsmi_get_clock_info" else (h, 0) if name == "amdsmi_get_power_cap_info" else (h,)
            print("==", name, json.dumps(_try(f, *a), default=str)[:6000])
    recs = [sample(amdsmi, h)]
    proc = subprocess.Popen(cmd) if cmd else None
    t_end = time.time() + (1e9 if proc else 3.0)
    while time.time() < t_end and (proc is None or proc.poll() is None):
        time.sleep(period)
        recs.append(sample(amdsmi, h))
    if dump:
        for r in recs[:: max(1, len(recs) // 12)]:
            print(json.dumps(r, default=str)[:3000])
    print(json.dumps(summarise(recs)))


if __name__ == "__main__":
    main()
