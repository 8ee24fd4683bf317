"""Aggregate the per-launch lines of USE_HIP_PROFILE_VERBOSE=1 (stderr of gpu_time_forward.py) by conv shape."""
import re, sys, collections
agg = collections.OrderedDict(); tot = 0.0
for l in open(sys.argv[1]):
    m = re.search(r'H=\s*(\d+) W=\s*(\d+) Cin=\s*(\d+) Cout=\s*(\d+) taps=(\d) gn=(\d) res=(\d) sc=(\d+)\s+([\d.]+) ms\s+([\d.]+) TF', l)
    if not m: continue
    k = tuple(int(x) for x in m.groups()[:8]); ms = float(m.group(9))
    a = agg.setdefault(k, [0, 0.0]); a[0] += 1; a[1] += ms; tot += ms
for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    H, W, ci, co, t, gn, res, sc = k
    fl = 2 * 8 * H * W * co * (ci * t + sc) * n
    print(f"H={H:4d} Cin={ci:4d} Cout={co:4d} gn={gn} res={res} sc={sc:4d}  n={n:2d}  {ms:7.3f} ms  {fl/ms/1e9:7.1f} TF")
print("total", round(tot, 3))
