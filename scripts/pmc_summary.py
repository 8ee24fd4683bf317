#!/usr/bin/env python3
"""Aggregate rocprofv3 counter-collection CSVs (one per --pmc pass) per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [kernel-substring]"""
import csv, glob, os, sys
from collections import defaultdict

d = sys.argv[1]; filt = sys.argv[2] if len(sys.argv) > 2 else ""
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        if filt and filt not in k:
            continue
        key = (k[:90], row.get("Grid_Size", ""), row.get("LDS_Block_Size", ""))
        a = acc[key][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
for key in sorted(acc, key=lambda k: -acc[k].get("SQ_WAVE_CYCLES", [0, 1])[0]):
    n = max(v[1] for v in acc[key].values())
    print(f"\n{key[0]}  grid={key[1]} lds={key[2]} dispatches={n}")
    for c, (s, cnt) in sorted(acc[key].items()):
        print(f"    {c:28s} {s / cnt:16.1f}")
