"""Bring-up: error pattern of conv_v7 (variant 8) against conv_v4 (variant 4) on small maps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from scripts.gpu_conv_bench import run

def show(name, case, B=1):
    a = run(case, 4, 1, B, 1); b = run(case, 8, 1, B, 1)
    if a is None or b is None:
        print(name, "n/a", a is None, b is None); return
    d = np.abs(a[0] - b[0]); m = np.abs(a[0]).max()
    print(f"{name}: maxdiff {d.max():.4g} of {m:.4g}; stats rel {np.abs(a[1]-b[1]).max()/max(1e-30,np.abs(a[1]).max()):.3g}")
    if d.max() > 0:
        bad = d > 1e-6 * m
        print("  bad frac", bad.mean(), " by item", bad.mean(axis=(1,2,3)))
        print("  bad rows (y):", np.nonzero(bad.any(axis=(0,2,3)))[0][:40])
        print("  bad cols (x):", np.nonzero(bad.any(axis=(0,1,3)))[0][:40])
        print("  bad chans   :", np.nonzero(bad.any(axis=(0,1,2)))[0][:40])
        y, x, c = [int(v[0]) for v in np.nonzero(bad[0])] if bad[0].any() else (0, 0, 0)
        print("  first bad (y,x,c)", y, x, c, "v4", a[0][0, y, x, c:c+8], "v7", b[0][0, y, x, c:c+8])

# (H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res)
show("1 tile plain", (16, 32, 128, 0, 128, 0, 0, 0, 0, 0, 0))
show("1 tile temb ", (16, 32, 128, 0, 128, 0, 0, 0, 0, 1, 0))
show("1 tile gnact", (16, 32, 128, 0, 128, 0, 0, 1, 1, 1, 0))
show("1 tile res  ", (16, 32, 128, 0, 128, 0, 0, 1, 1, 1, 1))
show("2x2 tiles   ", (32, 64, 128, 0, 128, 0, 0, 1, 1, 1, 1))
show("2x2 nb2     ", (32, 64, 128, 0, 256, 0, 0, 1, 1, 1, 1))
show("4x4 cat B3  ", (64, 128, 128, 64, 128, 0, 0, 1, 1, 1, 1), B=3)
show("1 tile sc128", (16, 32, 128, 0, 128, 128, 0, 1, 1, 1, 0))
show("2x2 sc cat  ", (32, 64, 128, 0, 128, 128, 128, 1, 1, 0, 0))
show("4x4 sc384 B3", (64, 128, 128, 0, 128, 256, 128, 1, 1, 0, 0), B=3)
show("2x2 nb2 sc  ", (32, 64, 128, 128, 256, 256, 0, 1, 1, 1, 0), B=2)
