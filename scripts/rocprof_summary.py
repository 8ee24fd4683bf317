#!/usr/bin/env python3
"""Summarise `rocprofv3 --kernel-trace --stats --output-format csv` output (scripts/run_profile.sh) per kernel.
usage: rocprof_summary.py <dir-with-*_kernel_stats.csv> [bench-json-line-file] [profiled command]"""
import csv, glob, os, sys

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True))[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
# the profiled command: given, else by the directory name profile.sh uses (prof_roofline = one launch on the chip at a time, the run
# roofline.avg_launch_ms must agree with; prof_bench = the benchmark command itself, sub-batch streams overlapping)
cmd = sys.argv[3] if len(sys.argv) > 3 else ("python bench.py --roofline-only" if "roofline" in os.path.basename(os.path.normpath(d))
                                             else "python bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-secondary")
print(f"# rocprofv3 --kernel-trace --stats of `{cmd}` ({os.path.basename(f)})")
print(f"# total kernel time {tot / 1e6:.2f} ms over {sum(int(r['Calls']) for r in rows)} launches")
print(f"{'total ms':>10} {'%':>6} {'calls':>7} {'avg us':>9} {'min us':>8} {'max us':>8}  kernel")
for r in rows[:24]:
    print(f"{float(r['TotalDurationNs']) / 1e6:10.2f} {float(r['Percentage']):6.2f} {r['Calls']:>7} {float(r['AverageNs']) / 1e3:9.1f} "
          f"{float(r['MinNs']) / 1e3:8.1f} {float(r['MaxNs']) / 1e3:8.1f}  {r['Name'][:110]}")
if len(sys.argv) > 2 and os.path.exists(sys.argv[2]):
    for line in open(sys.argv[2]):
        if line.startswith('{"metric'):
            print("# bench line of the profiled run:", line.strip()[:600])
