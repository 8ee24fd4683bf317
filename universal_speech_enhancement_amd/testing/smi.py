"""Socket power / clock / limiter sampling through amdsmi (the GPU box's SMI library): what holds the shader clock below its maximum while
a workload runs.  Used by bench.py's power_probe (in-process sampling thread) and by scripts/smi_probe.py (around a command).

The violation accumulators (amdsmi_get_violation_status) count, at ~1 kHz, the time each limiter was active: PPT (socket power limit),
socket / VR / HBM thermal, PROCHOT.  The share of a window in which a limiter was active is delta(acc_x) / delta(acc_counter)."""
import json
import time


def _bdf_key(bdf: str):
    """'0000:05:00.0' / '0000:05:00' / '5:0.0' -> (domain, bus, device) for comparison across the HIP and amdsmi spellings."""
    import re
    m = re.search(r"(?:([0-9a-fA-F]{1,8}):)?([0-9a-fA-F]{1,2}):([0-9a-fA-F]{1,2})(?:\.([0-7]))?$", bdf.strip())
    if not m:
        return None
    return (int(m.group(1) or "0", 16), int(m.group(2), 16), int(m.group(3), 16))


def smi_open(pci_bus_id: str = None):
    """(amdsmi module, processor handle).  ``pci_bus_id`` (hipDeviceGetPCIBusId of the device the workload runs on) selects the handle
    whose BDF matches - with HIP_VISIBLE_DEVICES or a rank > 0 the HIP device index is not the amdsmi index; without it,
    or without a match, handle 0."""
    import amdsmi
    amdsmi.amdsmi_init()
    hs = amdsmi.amdsmi_get_processor_handles()
    want = _bdf_key(pci_bus_id) if pci_bus_id else None
    if want is not None:
        for h in hs:
            try:
                if _bdf_key(str(amdsmi.amdsmi_get_gpu_device_bdf(h))) == want:
                    return amdsmi, h
            except Exception:  # noqa: BLE001
                continue
    return amdsmi, hs[0]


def power_cap_watts(amdsmi, h):
    """The socket power cap the box enforces (amdsmi_get_power_cap_info: microwatts on this driver), or None."""
    try:
        c = amdsmi.amdsmi_get_power_cap_info(h)
        v = float(c.get("power_cap", 0))
        return round(v / 1e6 if v > 1e5 else v, 1) if v > 0 else None
    except Exception:  # noqa: BLE001
        return None


def _try(f, *a):
    try:
        return f(*a)
    except Exception as e:  # noqa: BLE001 (bring-up tool: report whatever the driver refuses)
        return {"error": str(e)[:120]}


def sample(amdsmi, h):
    """One record: violation accumulators + power / clock / temperature (whatever this driver reports)."""
    rec = {"t": time.time()}
    v = _try(amdsmi.amdsmi_get_violation_status, h)
    rec["viol"] = v
    p = _try(amdsmi.amdsmi_get_power_info, h)
    rec["power"] = p
    m = _try(amdsmi.amdsmi_get_gpu_metrics_info, h)
    if isinstance(m, dict) and "error" not in m:
        keep = ("throttle_status", "indep_throttle_status", "average_socket_power", "current_socket_power", "temperature_hotspot",
                "temperature_mem", "temperature_vrsoc", "current_gfxclk", "average_gfxclk_frequency", "current_gfxclks", "average_gfx_activity",
                "average_umc_activity", "accumulation_counter", "prochot_residency_acc", "ppt_residency_acc", "socket_thm_residency_acc",
                "vr_thm_residency_acc", "hbm_thm_residency_acc", "gfx_below_host_limit_ppt_acc", "gfx_below_host_limit_thm_acc",
                "gfx_below_host_limit_total_acc", "gfx_low_utilization_acc", "energy_accumulator")
        rec["metrics"] = {k: m[k] for k in keep if k in m}
        # the per-XCP statistics (gfx950: 8 XCDs in one partition) carry the per-XCD below-host-limit accumulators on some drivers
        if "xcp_stats" in m:
            rec["xcp_keys"] = sorted(m["xcp_stats"][0].keys()) if m["xcp_stats"] else []
    else:
        rec["metrics"] = m
    return rec


def summarise(recs):
    """Share of the window each limiter was active: accumulator deltas over the samples at >= 90 % of the peak socket power."""
    def power_of(r):
        p = r.get("power") or {}
        for k in ("current_socket_power", "average_socket_power", "socket_power"):
            v = p.get(k) if isinstance(p, dict) else None
            if isinstance(v, (int, float)) and v > 0:
                return float(v)
        m = r.get("metrics") or {}
        for k in ("current_socket_power", "average_socket_power"):
            v = m.get(k) if isinstance(m, dict) else None
            if isinstance(v, (int, float)) and v > 0:
                return float(v)
        return 0.0
    pw = [power_of(r) for r in recs]
    if not pw or max(pw) <= 0:
        return {"samples": len(recs), "error": "no power readings"}
    hot = [i for i, w in enumerate(pw) if w >= 0.9 * max(pw)]
    i0, i1 = hot[0], hot[-1]
    out = {"samples": len(hot), "socket_W": round(sum(pw[i] for i in hot) / len(hot), 1)}
    v0, v1 = recs[i0].get("viol"), recs[i1].get("viol")
    if isinstance(v0, dict) and isinstance(v1, dict) and "error" not in v0 and "error" not in v1 and i1 > i0:
        dc = None
        try:
            dc = float(v1["acc_counter"]) - float(v0["acc_counter"])
        except Exception:  # noqa: BLE001
            pass
        lim = {}
        for k in ("acc_ppt_pwr", "acc_socket_thrm", "acc_vr_thrm", "acc_hbm_thrm", "acc_prochot_thrm", "acc_gfx_clk_below_host_limit"):
            try:
                d = float(v1[k]) - float(v0[k])
                lim[k[4:]] = round(d / dc, 4) if dc else d
            except Exception:  # noqa: BLE001
                lim[k[4:]] = None
        for k in ("acc_gfx_clk_below_host_limit_pwr", "acc_gfx_clk_below_host_limit_thm", "acc_gfx_clk_below_host_limit_total", "acc_low_utilization"):
            try:
                a0, a1 = v0[k], v1[k]
                flat0 = [x for row in a0 for x in (row if isinstance(row, (list, tuple)) else [row])]
                flat1 = [x for row in a1 for x in (row if isinstance(row, (list, tuple)) else [row])]
                ds = [float(b) - float(a) for a, b in zip(flat0, flat1) if isinstance(a, (int, float)) and isinstance(b, (int, float)) and b < 2 ** 63 and a < 2 ** 63]
                ds = [d for d in ds if d >= 0]
                lim[k[4:] + "_per_xcd"] = [round(d / dc, 4) if dc else d for d in ds[:8]]
            except Exception:  # noqa: BLE001
                lim[k[4:] + "_per_xcd"] = None
        out["limit_reasons"] = lim
        out["window_s"] = round(recs[i1]["t"] - recs[i0]["t"], 2)
    ms = [recs[i].get("metrics") for i in hot if isinstance(recs[i].get("metrics"), dict)]
    for k in ("temperature_hotspot", "temperature_mem", "current_gfxclk", "average_gfxclk_frequency"):
        vals = [m[k] for m in ms if isinstance(m.get(k), (int, float)) and m[k] < 65535]
        if vals:
            out[k] = round(sum(vals) / len(vals), 1)
    return out
