"""Bring-up of a convolution variant (argv: batch, variant; default 10): compare with the generic kernel on small shapes and print where
the outputs differ."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import importlib.util
spec = importlib.util.spec_from_file_location("gcb", os.path.join(os.path.dirname(os.path.abspath(__file__)), "gpu_conv_bench.py"))
g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
cases = {
    "1 tile":      (8, 32, 128, 0, 128, 0, 0, 1, 1, 1, 0),
    "2x2 tiles":   (16, 64, 128, 0, 128, 0, 0, 1, 1, 1, 0),
    "4x3 +res":    (32, 96, 128, 0, 128, 0, 0, 1, 1, 0, 1),
    "cat 256+128": (16, 64, 256, 128, 128, 0, 0, 1, 1, 1, 0),
    "cout 256":    (16, 64, 256, 0, 256, 0, 0, 1, 1, 1, 0),
}
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
VAR = int(sys.argv[2]) if len(sys.argv) > 2 else 10
for name, case in cases.items():
    ref = g.run(case, 1, 1, B, 1)
    got = g.run(case, VAR, 1, B, 1)
    v4 = g.run(case, 4, 1, B, 1)
    if v4 is not None:
        d4 = np.abs(v4[0] - ref[0]); print(f"   [v4 vs generic: maxdiff {d4.max():.3g}, differing {np.mean(d4 > 0):.4f}; v9 vs generic differing {np.mean(np.abs(got[0] - ref[0]) > 0):.4f}; v9 vs v4 maxdiff {np.abs(got[0] - v4[0]).max():.3g} differing {np.mean(np.abs(got[0] - v4[0]) > 0):.4f}]")
    if got is None:
        print(name, "n/a"); continue
    d = np.abs(got[0] - ref[0]); m = np.abs(ref[0]).max()
    bad = ~np.isfinite(got[0]) | (d > 0.02 * m)
    ds = np.abs(got[1] - ref[1]).max() / max(1e-30, np.abs(ref[1]).max())
    print(f"{name:12s} B={B} maxdiff {np.nanmax(d):.3g} of {m:.3g}  bad {bad.mean():.4f}  nan {np.isnan(got[0]).mean():.4f}  stats rel {ds:.3g}")
    if bad.any():
        bb, yy, xx, cc = np.nonzero(bad)
        print("   bad items", np.unique(bb), "rows", np.unique(yy)[:40], "cols", np.unique(xx)[:40], "ch", np.unique(cc)[:40])
        i = (bb[0], yy[0], xx[0], cc[0]); print("   first", i, got[0][i], ref[0][i])
    big = d > 0.004 * m
    if big.any():
        bb, yy, xx, cc = np.nonzero(big)
        print("   >0.4%:", big.mean(), "rows", np.bincount(yy, minlength=case[0])[:16], "cols", np.bincount(xx, minlength=case[1])[:34], "ch/8", np.bincount(cc // 8, minlength=case[4] // 8))
