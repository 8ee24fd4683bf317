import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from scripts.gpu_conv_bench import run
dt = int(sys.argv[1]) if len(sys.argv) > 1 else 0
va, vb = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (2, 4)
def cmp(name, case, B):
    a = run(case, va, 1, B, dt); b = run(case, vb, 1, B, dt)
    d = np.abs(a[0] - b[0]); bad = d > 0.05
    print(f"{name}: maxdiff {d.max():.3g}, bad {bad.sum()} of {bad.size}")
    if bad.any():
        idx = np.argwhere(bad)
        print("  bad b:", np.unique(idx[:, 0]), "\n  rows:", np.unique(idx[:, 1])[:48], "\n  cols:", np.unique(idx[:, 2])[:70], "\n  ch:", np.unique(idx[:, 3])[:130])
        y, x = idx[0][1], idx[0][2]
        print("  sample ref:", a[0][idx[0][0], y, x, :12], "\n  got:", b[0][idx[0][0], y, x, :12])
cmp("16x32 C128", (16, 32, 128, 0, 128, 0, 0, 1, 1, 1, 0), 1)
cmp("32x64 C128", (32, 64, 128, 0, 128, 0, 0, 1, 1, 1, 0), 1)
