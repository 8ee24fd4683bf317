import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from scripts.gpu_conv_bench import run
from universal_speech_enhancement_amd import _lib
from universal_speech_enhancement_amd._lib import check
for o in sys.argv[1:]:
    k, v = o.split("="); check(_lib.lib().use_set_option(k.encode(), int(v)))
def cmp(name, case, B):
    a = run(case, 4, 1, B, 1); b = run(case, 5, 1, B, 1)
    d = np.abs(a[0] - b[0]); bad = d > 0.05
    print(f"{name}: maxdiff {d.max():.3g}, bad {bad.sum()} of {bad.size}")
    if bad.any():
        idx = np.argwhere(bad)
        print("  bad b:", np.unique(idx[:, 0]), " rows:", np.unique(idx[:, 1])[:40], " cols:", np.unique(idx[:, 2])[:40], " ch:", np.unique(idx[:, 3])[:40])
        # per-tile count
        t = {}
        for bb, y, x, c in idx[:20000]:
            t[(bb, y // 8, x // 32)] = t.get((bb, y // 8, x // 32), 0) + 1
        print("  tiles:", sorted(t.items())[:30], len(t))
#              H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res
cmp("small 16x64 C128", (16, 64, 128, 0, 128, 0, 0, 1, 1, 1, 0), 1)
cmp("small 64x64 C128", (64, 64, 128, 0, 128, 0, 0, 1, 1, 1, 0), 2)
cmp("mid 256x320 C128", (256, 320, 128, 0, 128, 0, 0, 1, 1, 1, 0), 4)
cmp("mid 256x320 C128 noact", (256, 320, 128, 0, 128, 0, 0, 0, 0, 0, 0), 4)
cmp("L0 512x640 C128", (512, 640, 128, 0, 128, 0, 0, 1, 1, 1, 0), 4)
cmp("L0 512x640 C128 B1", (512, 640, 128, 0, 128, 0, 0, 1, 1, 1, 0), 1)
