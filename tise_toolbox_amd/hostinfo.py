"""What the host really gives this process (no torch import: the decode workers' parent and the tests use it early)."""
import os


def usable_cpus():
    """CPUs this process may really use: the affinity mask AND the cgroup's CFS quota.  The GPU boxes show 256 hardware
    threads and give the container ``cpu.max = 1600000 100000`` -- 16 CPUs of time per period; decode processes beyond the
    quota are throttled in bursts and the feed gets SLOWER (12.5 k images/s with 16 workers, 8.7 k with 64, 5.9 k with 128:
    profiles/r05b_host_decode_probe.txt; round 4's DataLoader feed fell the same way between 32 and 64 workers)."""
    n = os.cpu_count() or 8
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        pass
    for path in ("/sys/fs/cgroup/cpu.max",):                                   # cgroup v2
        try:
            quota, period = open(path).read().split()[:2]
            if quota != "max":
                n = min(n, max(1, -(-int(quota) // int(period))))
        except (OSError, ValueError):
            pass
    try:                                                                       # cgroup v1
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0 and per > 0:
            n = min(n, max(1, -(-q // per)))
    except (OSError, ValueError):
        pass
    return n


def limit_torch_threads():
    """Keep torch's CPU thread pool inside the CPUs the process may really use.  On the GPU boxes torch sees 256 hardware
    threads while the container's cgroup grants 16 CPUs of time: every CPU operator (BatchNorm folding, weight packing, the
    tokenisers' index work) then wakes a 256-thread OpenMP team that the CFS quota throttles in bursts -- seconds of start-up
    per process.  An explicit OMP_NUM_THREADS / torch.set_num_threads by the user is respected (never raised)."""
    import torch
    n = usable_cpus()
    if torch.get_num_threads() > n:
        torch.set_num_threads(n)
    return torch.get_num_threads()


def cfs_throttle():
    """(periods throttled, microseconds throttled) of this process's cgroup so far (cgroup v2 cpu.stat; (0, 0) when unknown): a
    container that runs more threads than its CPU quota gets ALL of them -- the thread that launches kernels included -- stopped
    for the rest of a 100 ms period."""
    try:
        d = dict(line.split() for line in open("/sys/fs/cgroup/cpu.stat"))
        return int(d.get("nr_throttled", 0)), int(d.get("throttled_usec", 0))
    except (OSError, ValueError):
        return 0, 0

