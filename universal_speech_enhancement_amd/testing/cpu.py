"""Host-core count for the CPU oracle runs of the test infrastructure (tests, smoke, bench.py's cpu_baseline)."""
import os


def usable_cores(cap: int = 64) -> int:
    """Cores this process may really use: affinity mask capped by the cgroup CPU quota (a box with 256 hardware threads and a
    16-core quota runs a 256-thread torch 20x slower than a 16-thread one)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max",):
        try:
            quota, period = open(path).read().split()
            if quota != "max":
                n = min(n, max(1, int(float(quota) / float(period))))
        except Exception:
            pass
    try:                                                    # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read()); p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            n = min(n, max(1, q // p))
    except Exception:
        pass
    return max(1, min(cap, n))
