#!/usr/bin/env python3
"""Summarise the rocprofv3 --pmc passes of scripts/profile.sh pmc for one kernel into the JSON that bench.py quotes.
usage: pmc_to_json.py <pmc-dir> <kernel-substring> <out.json>
HBM bytes follow MI355X_MICROARCH.md (HBM section): rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
FETCH_SIZE tallies 128-byte read requests at 64 B, so it is doubled."""
import csv, glob, json, os, sys
from collections import defaultdict

d, filt, out = sys.argv[1], sys.argv[2], sys.argv[3]
acc = defaultdict(lambda: [0.0, 0])
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(f)):
        if filt not in row["Kernel_Name"]:
            continue
        a = acc[row["Counter_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
avg = {k: v[0] / v[1] for k, v in acc.items() if v[1]}
n = max(v[1] for v in acc.values())
rd = avg.get("FETCH_SIZE", 0.0) * 1024.0 * 2.0          # KB -> B, x2: gfx950 correction (see the guide's HBM section)
wr = avg.get("WRITE_SIZE", 0.0) * 1024.0
res = {
    "kernel": filt, "workload": "one score evaluation at configs[1] (B=8, T'=640), scripts/profile.sh pmc",
    "dispatches_averaged": n,
    "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
    "fetch_size_correction": "x2 (gfx950 FETCH_SIZE = TCC_EA0_RDREQ x 64 B for 128-B requests)",
}
g = avg.get("GRBM_GUI_ACTIVE")          # summed over the 8 XCDs: x128 = SIMD-cycles, x64 / x32 = per-CU pair / LDS units
if g:
    if "SQ_VALU_MFMA_BUSY_CYCLES" in avg: res["mfma_busy_frac"] = avg["SQ_VALU_MFMA_BUSY_CYCLES"] / (g * 128.0)
    if "SQ_ACTIVE_INST_VALU" in avg: res["valu_active_frac"] = avg["SQ_ACTIVE_INST_VALU"] / (g * 64.0)
    if "SQ_LDS_IDX_ACTIVE" in avg: res["lds_active_frac"] = avg["SQ_LDS_IDX_ACTIVE"] / (g * 32.0)
if "SQ_WAVE_CYCLES" in avg:
    for k, name in (("SQ_WAIT_ANY", "wave_wait_any_frac"), ("SQ_WAIT_INST_ANY", "wave_wait_inst_frac")):
        if k in avg:
            res[name] = avg[k] / avg["SQ_WAVE_CYCLES"]
if "TCC_HIT_sum" in avg and "TCC_MISS_sum" in avg:
    res["l2_hit_rate"] = avg["TCC_HIT_sum"] / (avg["TCC_HIT_sum"] + avg["TCC_MISS_sum"])
if "SQ_LDS_BANK_CONFLICT" in avg and "SQ_LDS_IDX_ACTIVE" in avg:
    res["lds_bank_conflict_frac_of_lds"] = avg["SQ_LDS_BANK_CONFLICT"] / avg["SQ_LDS_IDX_ACTIVE"]
# effective shader clock while the kernel runs: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / the kernel's duration in the kernel trace of
# the SAME pass (PMC passes serialise dispatches, so the duration is the kernel alone on the chip)
gtot, ttot = 0.0, 0.0
for f in sorted(glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if filt in r["Kernel_Name"] and r["Counter_Name"] == "GRBM_GUI_ACTIVE"]
    if not rows:
        continue
    tr = f.replace("counter_collection.csv", "kernel_trace.csv")
    if not os.path.exists(tr):
        continue
    dur = {r["Dispatch_Id"]: float(r["End_Timestamp"]) - float(r["Start_Timestamp"]) for r in csv.DictReader(open(tr)) if filt in r["Kernel_Name"]}
    for r in rows:
        if r["Dispatch_Id"] in dur:
            gtot += float(r["Counter_Value"]); ttot += dur[r["Dispatch_Id"]]
if ttot > 0:
    res["effective_clock_GHz"] = gtot / 8.0 / ttot
    res["effective_clock_note"] = "GRBM_GUI_ACTIVE / 8 XCDs / kernel-trace duration of the same (counter) pass; dispatches are serialised under --pmc"
# provenance: bench.py quotes hbm_bytes_per_launch only while the kernel source is the one these counters were collected on
import hashlib
_file = "use_conv_v5.hip" if "conv_v5" in filt else "use_conv_v4.hip"          # the kernel's source file, by the kernel-name filter
_src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "universal_speech_enhancement_amd", "csrc", _file)
res["kernel_source"] = "universal_speech_enhancement_amd/csrc/" + _file
res["kernel_source_sha16"] = hashlib.sha256(open(_src, "rb").read()).hexdigest()[:16]
res["raw_avg_per_dispatch"] = avg
json.dump(res, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if k != "raw_avg_per_dispatch"}, indent=1))
