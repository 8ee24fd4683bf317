#!/usr/bin/env python3
"""Timeline analysis of a rocprofv3 kernel trace of graph-replayed score evaluations: idle time (no kernel on the chip), time with
only under-filling kernels resident (< 256 workgroups in total), per-kernel-class exposed time.  usage: trace_gaps.py <kernel_trace.csv>"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    wg = (int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    n = r['Kernel_Name']
    cls = ('v4' if 'conv_v4' in n else 'v2' if 'conv_v2' in n else 'sk' if 'conv_sk' in n else 'pyr' if 'pyr_conv' in n else
           'conv_in' if 'conv_in' in n else 'gen' if 'conv_kernel' in n else 'fir' if 'fir_' in n else 'gn' if 'gn_finalize' in n else
           'attn' if 'attention' in n else 'sde' if any(k in n for k in ('predictor', 'corrector', 'langevin', 'prior')) else 'other')
    ev.append((s, e, wg, cls))
ev.sort()
# window = the last complete sampler call: from its prior_kernel to its last predictor_kernel
priors = [int(r['Start_Timestamp']) for r in rows if 'prior_kernel' in r['Kernel_Name']]
preds = [int(r['End_Timestamp']) for r in rows if 'predictor_kernel' in r['Kernel_Name']]
if priors and preds:
    t0 = max(priors); t1 = max(preds)
    ev = [x for x in ev if x[0] >= t0 and x[1] <= t1]
pts = []
for s, e, wg, c in ev:
    pts.append((s, 1, wg, c)); pts.append((e, -1, wg, c))
pts.sort()
active = collections.Counter(); wgs = 0; last = pts[0][0]
idle = under = full = 0
alone = collections.Counter()
for t, d, wg, c in pts:
    dt = t - last
    if dt > 0:
        n = sum(active.values())
        if n == 0: idle += dt
        elif wgs < 256:
            under += dt
            alone['+'.join(sorted(k for k, v in active.items() if v))] += dt
        else: full += dt
    active[c] += d; wgs += d * wg; last = t
tot = idle + under + full
print(f"window {tot/1e6:.2f} ms: idle {idle/tot:.3f}  under-filled(<256 WGs resident) {under/tot:.3f}  filled {full/tot:.3f}")
for k, v in alone.most_common(12):
    print(f"   under-filled with only [{k}] resident: {v/tot:.3f}")
# idle intervals: by (kernel class that ended, kernel class that starts)
ends = sorted((e, c) for s, e, wg, c in ev); starts = sorted((s, c) for s, e, wg, c in ev)
import bisect
cur_end = ev[0][0]; gaps = collections.Counter(); gcount = collections.Counter(); hist = collections.Counter()
evs = sorted(ev)
maxend = evs[0][1]; lastc = evs[0][3]
for s, e, wg, c in evs[1:]:
    if s > maxend:
        g = s - maxend
        gaps[(lastc, c)] += g; gcount[(lastc, c)] += 1
        hist[min(int(g / 1000), 20)] += g
    if e > maxend: maxend = e; lastc = c
print("idle by (ended, started) class [ms, count, avg us]:")
for k, v in gaps.most_common(14):
    print(f"   {k[0]:>8} -> {k[1]:<8} {v/1e6:8.2f} ms  n={gcount[k]:5d}  avg {v/gcount[k]/1e3:6.1f} us")
print("idle time by gap length (us bucket: share):", {k: round(v / max(1, sum(hist.values())), 3) for k, v in sorted(hist.items())})
