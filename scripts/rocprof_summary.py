#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (``--kernel-trace --stats``) as a per-kernel table (top_kernels view)."""
import sqlite3
import sys

db, out = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else None)
con = sqlite3.connect(db)
rows = list(con.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
lines = [f"{'calls':>8} {'total_ms':>12} {'avg_us':>10} {'pct':>7}  kernel"]
for name, calls, tot, avg, pct in rows:
    if len(name) > 150:
        name = name[:147] + "..."
    lines.append(f"{calls:8d} {tot / 1e3:12.3f} {avg:10.3f} {pct:7.3f}  {name}")
text = "\n".join(lines) + "\n"
if out:
    open(out, "a").write(text)
else:
    print(text)
