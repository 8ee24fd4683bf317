"""Single-convolution A/B harness on the GPU box (use_conv_bench): times conv_v2 (2) / conv_v4 (4) / conv_sk (7) / the generic kernel (1) on the layer shapes of the
NCSN++ Large score network at the configs[1] sub-batch (B=4) and checks that the variants agree.

    python scripts/gpu_conv_bench.py [--variants 4,2] [--iters 10] [--cases main|all] [--opt name=value ...]
"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from universal_speech_enhancement_amd import _lib  # noqa: E402
from universal_speech_enhancement_amd._lib import UseConvCase, check  # noqa: E402

# name: (H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res)
CASES = {
    "L0 conv0 128->128":        (512, 640, 128, 0, 128, 0, 0, 1, 1, 1, 0),
    "L0 conv1 128->128 +res":   (512, 640, 128, 0, 128, 0, 0, 1, 1, 0, 1),
    "L0 conv0 cat256->128":     (512, 640, 128, 128, 128, 0, 0, 1, 1, 1, 0),
    "L0 conv1 128->128 +sc256": (512, 640, 128, 0, 128, 128, 128, 1, 1, 0, 0),
    "L1 conv0 cat384->128":     (256, 320, 256, 128, 128, 0, 0, 1, 1, 1, 0),
    "L1 conv1 128->128 +sc384": (256, 320, 128, 0, 128, 256, 128, 1, 1, 0, 0),
    "L1 conv0 128->128 noact":  (256, 320, 128, 0, 128, 0, 0, 0, 0, 1, 0),
    "L1 conv0 128->128":        (256, 320, 128, 0, 128, 0, 0, 1, 1, 1, 0),
    "L1 conv1 128->128 +res":   (256, 320, 128, 0, 128, 0, 0, 1, 1, 0, 1),
    "L1 conv0 256->256":        (256, 320, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L2 conv0 256->256":        (128, 160, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L2 conv1 256->256 +res":   (128, 160, 256, 0, 256, 0, 0, 1, 1, 0, 1),
    "L2 conv0 cat512->256":     (128, 160, 256, 256, 256, 0, 0, 1, 1, 1, 0),
    "L3 conv0 256->256":        (64, 80, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L3 conv0 cat512->256":     (64, 80, 256, 256, 256, 0, 0, 1, 1, 1, 0),
    "L4 conv0 256->256":        (32, 40, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L4 conv1 256->256 +res":   (32, 40, 256, 0, 256, 0, 0, 1, 1, 0, 1),
    "L4 conv0 cat512->256":     (32, 40, 256, 256, 256, 0, 0, 1, 1, 1, 0),
    "L4 conv1 256->256 +sc512": (32, 40, 256, 0, 256, 256, 256, 1, 1, 0, 0),
    "L5 conv0 256->256":        (16, 20, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L5 conv0 cat512->256":     (16, 20, 256, 256, 256, 0, 0, 1, 1, 1, 0),
    "L5 conv1 256->256 +sc512": (16, 20, 256, 0, 256, 256, 256, 1, 1, 0, 0),
    "L6 conv0 256->256":        (8, 10, 256, 0, 256, 0, 0, 1, 1, 1, 0),
    "L6 conv1 256->256 +res":   (8, 10, 256, 0, 256, 0, 0, 1, 1, 0, 1),
    "L6 conv0 cat512->256":     (8, 10, 256, 256, 256, 0, 0, 1, 1, 1, 0),
    "L6 conv0 noact nogn":      (8, 10, 256, 0, 256, 0, 0, 0, 0, 1, 0),
    "odd 7x9 96->96 +sc192":    (7, 9, 96, 0, 96, 96, 96, 1, 1, 0, 0),
    "odd 5x33 cat160->64":      (5, 33, 96, 64, 64, 0, 0, 1, 1, 1, 0),
}
MAIN = ["L0 conv0 128->128", "L0 conv1 128->128 +res", "L0 conv1 128->128 +sc256", "L1 conv0 cat384->128", "L2 conv0 256->256"]


def run(case, variant, iters, B, dtype, want_out=True):
    H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res = case
    c = UseConvCase(B, H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res, 1, dtype, variant, iters)
    out = np.empty((B, H, W, Cout), np.float32) if want_out else None
    st = np.empty((B, Cout, 2), np.float32)
    ms, fl = C.c_double(), C.c_double()
    rc = _lib.lib().use_conv_bench(C.byref(c), None if out is None else out.ctypes.data_as(C.c_void_p), st.ctypes.data_as(C.c_void_p),
                                   C.byref(ms), C.byref(fl))
    if rc:
        return None
    return out, st, ms.value, fl.value


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="4,2")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--cases", default="main")
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--dtype", type=int, default=1)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--opt", action="append", default=[], help="use_set_option name=value (repeatable)")
    ap.add_argument("--no-check", action="store_true")
    a = ap.parse_args()
    for o in a.opt:
        k, v = o.split("=")
        check(_lib.lib().use_set_option(k.encode(), int(v)), "use_set_option")
    variants = [int(v) for v in a.variants.split(",")]
    names = (MAIN if a.cases == "main" else list(CASES) if a.cases == "all" else
             [n for n in CASES if any(k in n for k in a.cases.split(","))])
    for name in names:
        case = CASES[name]
        ref = None
        line = f"{name:28s}"
        for v in variants:
            best = None
            for r in range(a.rounds):
                got = run(case, v, a.iters, a.batch, a.dtype, want_out=(r == 0 and not a.no_check))
                if got is None:
                    break
                best = got[2] if best is None else min(best, got[2])
                if r == 0:
                    first = got
            if best is None:
                line += f" | v{v}: n/a"
                continue
            line += f" | v{v}: {best:7.3f} ms {first[3] / best / 1e9:7.1f} TF"
            if not a.no_check:
                if ref is None:
                    ref = first
                else:
                    d = float(np.abs(first[0] - ref[0]).max()); m = float(np.abs(ref[0]).max())
                    ds = float(np.abs(first[1] - ref[1]).max() / max(1e-30, np.abs(ref[1]).max()))
                    line += f" (maxdiff {d:.3g} of {m:.3g}, stats rel {ds:.2g}, finite {bool(np.isfinite(first[0]).all())})"
        print(line, flush=True)


if __name__ == "__main__":
    main()
