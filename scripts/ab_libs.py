"""Same-box A/B of library builds (USE_HIP_LIB): single-convolution harness timings, bit-identity of the outputs across builds, and the
end-to-end score evaluation.  Builds alternate (A B A B) so that clock / thermal drift hits both.

    python scripts/ab_libs.py [--variant 4] [--cases main|all|substr,...] [--e2e] libA.so libB.so [...]
"""
import argparse
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r"""
import os, sys, json
sys.path.insert(0, os.environ["USE_ROOT"])
import numpy as np
sys.argv = ["x"]
import importlib.util
spec = importlib.util.spec_from_file_location("gcb", os.path.join(os.environ["USE_ROOT"], "scripts", "gpu_conv_bench.py"))
g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
names = json.loads(os.environ["AB_CASES"]); variant = int(os.environ["AB_VARIANT"]); iters = int(os.environ["AB_ITERS"])
from universal_speech_enhancement_amd import _lib
for kv in filter(None, os.environ.get("AB_OPTS", "").split(";")):
    _lib.check(_lib.lib().use_set_option(kv.split("=")[0].encode(), int(kv.split("=")[1])), "use_set_option")
res = {}
for n in names:
    best = None
    for r in range(3):
        got = g.run(g.CASES[n], variant, iters, 4, 1, want_out=(r == 0))
        if got is None: break
        if r == 0:
            import zlib
            res[n] = {"crc": zlib.crc32(got[0].tobytes()), "stats_crc": zlib.crc32(got[1].tobytes()), "finite": bool(np.isfinite(got[0]).all()), "flops": got[3]}
        best = got[2] if best is None else min(best, got[2])
    if best is not None: res[n]["ms"] = best
print("AB_RESULT " + json.dumps(res))
"""


def main():
    import json
    ap = argparse.ArgumentParser()
    ap.add_argument("libs", nargs="+")
    ap.add_argument("--variant", type=int, default=4)
    ap.add_argument("--cases", default="main")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--rounds", type=int, default=2)
    ap.add_argument("--e2e", action="store_true")
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    sys.argv = ["x"]
    import importlib.util
    spec = importlib.util.spec_from_file_location("gcb", os.path.join(ROOT, "scripts", "gpu_conv_bench.py"))
    g = importlib.util.module_from_spec(spec); spec.loader.exec_module(g)
    names = (g.MAIN if a.cases == "main" else list(g.CASES) if a.cases == "all" else [n for n in g.CASES if any(k in n for k in a.cases.split(","))])
    results = {}
    for rnd in range(a.rounds):
        for lib in a.libs:
            path, _, opts = lib.partition("@")            # lib.so@option=value;option=value
            env = dict(os.environ, USE_ROOT=ROOT, USE_HIP_LIB=os.path.join(ROOT, path), AB_CASES=json.dumps(names), AB_VARIANT=str(a.variant), AB_ITERS=str(a.iters), AB_OPTS=opts)
            r = subprocess.run([sys.executable, "-c", WORKER], env=env, capture_output=True, text=True, timeout=1200)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("AB_RESULT ")]
            if not line:
                print(lib, "FAILED", r.stdout[-1000:], r.stderr[-2000:]); continue
            results.setdefault(lib, []).append(json.loads(line[-1][10:]))
    for n in names:
        row = f"{n:28s}"
        crcs = set()
        for lib in a.libs:
            runs = [r[n] for r in results.get(lib, []) if n in r and "ms" in r[n]]
            if not runs:
                row += f" | {os.path.basename(lib)}: n/a"; continue
            ms = min(r["ms"] for r in runs)
            tag = (os.path.basename(lib.partition("@")[0])[10:-3] or "head") + ("@" + lib.partition("@")[2] if "@" in lib else "")
            row += f" | {tag}: {ms:7.3f} ms {runs[0]['flops'] / ms / 1e9:7.1f} TF"
            crcs.add((runs[0]["crc"], runs[0]["stats_crc"], runs[0]["finite"]))
        row += "  outputs " + ("IDENTICAL" if len(crcs) == 1 else f"DIFFER {crcs}")
        print(row, flush=True)
    if a.e2e:
        for rnd in range(2):
            for lib in a.libs:
                path, _, opts = lib.partition("@")
                env = dict(os.environ, USE_HIP_LIB=os.path.join(ROOT, path), USE_OPTS=opts.replace(";", ","))
                r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "gpu_time_forward.py"), "bf16", "8", "640", "10"], env=env, capture_output=True, text=True, timeout=1200)
                print(f"e2e {os.path.basename(lib):28s} {r.stdout.strip().splitlines()[-1][:110] if r.stdout.strip() else r.stderr[-500:]}", flush=True)


if __name__ == "__main__":
    main()
