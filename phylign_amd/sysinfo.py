"""What the host really grants this process (scheduler affinity and cgroup CPU quota): the GPU boxes show 256 logical
CPUs but give a job a 16-CPU quota, and threads beyond the quota are only throttled."""
import os

import numpy as np


def effective_cpus():
    """CPUs this process may really use: the scheduler affinity, capped by the cgroup CPU quota
    (the GPU box shows 256 logical CPUs but grants the job a 16-CPU quota: 256 threads would
    only be throttled)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]          # cgroup v2
        if quota != "max":
            n = min(n, max(1, int(np.ceil(int(quota) / int(period)))))
    except Exception:
        try:
            quota = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())          # cgroup v1
            period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if quota > 0:
                n = min(n, max(1, int(np.ceil(quota / period))))
        except Exception:
            pass
    return max(1, n)


def available_ram_gb():
    """GB of host RAM this process may still take: MemAvailable, capped by what the cgroup leaves"""
    avail = None
    try:
        for line in open("/proc/meminfo"):
            if line.startswith("MemAvailable:"):
                avail = int(line.split()[1]) * 1024
    except Exception:
        pass
    try:
        lim = open("/sys/fs/cgroup/memory.max").read().strip()
        if lim != "max":
            cur = int(open("/sys/fs/cgroup/memory.current").read())
            avail = min(avail, int(lim) - cur) if avail is not None else int(lim) - cur
    except Exception:
        pass
    return (avail or 8 << 30) / 1e9
