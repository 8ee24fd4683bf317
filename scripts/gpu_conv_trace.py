"""Bring-up: cycle-stamp timeline of one workgroup of conv_v7 / conv_v4 (trace build: USE_HIP_LIB=.../libuse_hip_trace.so,
USE_HIP_TRACE=<workgroup>).  usage: gpu_conv_trace.py <variant> <case name substring>"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from scripts.gpu_conv_bench import run, CASES
v = int(sys.argv[1]); key = sys.argv[2]
name = [n for n in CASES if key in n][0]
print(name, "variant", v, flush=True)
got = run(CASES[name], v, 3, 4, 1, want_out=False)
print("ms", got[2] if got else None)
