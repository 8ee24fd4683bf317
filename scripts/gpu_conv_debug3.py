import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import numpy as np
from scripts.gpu_conv_bench import run
def cmp(name, case, B=1):
    a = run(case, 0, 1, B, 0); b = run(case, 0, 1, B, 1)
    d = np.abs(a[0] - b[0]).max(); m = np.abs(a[0]).max()
    print(f"{name}: fp32 vs bf16 (dispatcher): maxdiff {d:.3g} of {m:.3g}  stats rel {np.abs(a[1]-b[1]).max()/np.abs(a[1]).max():.2g}")
#            H, W, C0, C1, Cout, XC0, XC1, act, gn, temb, res
cmp("C96->96 64x32", (64, 32, 96, 0, 96, 0, 0, 1, 1, 1, 0))
cmp("C96->192 64x32", (64, 32, 96, 0, 192, 0, 0, 1, 1, 1, 0))
cmp("C192->192 64x32", (64, 32, 192, 0, 192, 0, 0, 1, 1, 1, 0))
cmp("C288(192+96)->96", (64, 32, 192, 96, 96, 0, 0, 1, 1, 1, 0))
cmp("C96->96 +sc 288", (64, 32, 96, 0, 96, 192, 96, 1, 1, 0, 0))
cmp("C192->192 +sc 96", (64, 32, 192, 0, 192, 96, 0, 1, 1, 0, 0))
cmp("C96->96 512x64", (512, 64, 96, 0, 96, 0, 0, 1, 1, 1, 0))
cmp("C96->96 8x16 small", (8, 16, 96, 0, 96, 0, 0, 1, 1, 1, 0))
cmp("C128->128 64x32 (ref)", (64, 32, 128, 0, 128, 0, 0, 1, 1, 1, 0))
