"""Pick the host thread count for the CPU oracle (TEST INFRASTRUCTURE ONLY).

GPU boxes report hundreds of logical CPUs while the container may be throttled by a cgroup quota; running torch's
CPU backend with os.cpu_count() threads there is ~200x slower than with a sane count.  `configure()` bounds the
candidates by affinity and cgroup quota and keeps the fastest on a short matmul calibration."""
import os
import time

import torch


def _cgroup_cpus():
    try:
        q, p = open('/sys/fs/cgroup/cpu.max').read().split()
        if q != 'max':
            return max(1, int(int(q) / int(p)))
    except (OSError, ValueError):
        pass
    try:
        q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
        p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
        if q > 0:
            return max(1, q // p)
    except (OSError, ValueError):
        pass
    return None


def available():
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    c = _cgroup_cpus()
    return min(n, c) if c else n


def configure(max_threads=64):
    """set torch's intra-op threads to the fastest of {4, 8, 16, 32, 64, ...} <= available(); returns the count."""
    limit = min(available(), max_threads)
    cands = sorted({c for c in (4, 8, 16, 32, 64, limit) if c <= limit} or {1})
    a, b = torch.randn(1024, 1024), torch.randn(1024, 1024)
    best, best_t = cands[0], float('inf')
    for c in cands:
        torch.set_num_threads(c)
        a @ b
        t0 = time.perf_counter()
        for _ in range(3):
            a @ b
        dt = time.perf_counter() - t0
        if dt < best_t * 0.9:
            best, best_t = c, dt
    torch.set_num_threads(best)
    return best
