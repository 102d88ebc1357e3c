#!/usr/bin/env python3
"""What the host of a GPU box gives a PNG feed: cgroup CPU quota, per-thread decode time, and the decode rate of the
shared-ring workers (png_ring.PngRingLoader.iter_host: no GPU involved) at several worker counts."""
import os, sys, time, tempfile
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from concurrent.futures import ProcessPoolExecutor


def show(path):
    try:
        print(f"{path}: {open(path).read().strip()[:300]}")
    except OSError as e:
        print(f"{path}: {e}")


def write(args):
    d, i = args
    rng = np.random.default_rng(i)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    img = np.zeros((256, 256, 3), np.float32)
    for c in range(3):
        for _ in range(4):
            fx, fy, ph = rng.uniform(0.01, 0.15, 2).tolist() + [rng.uniform(0, 6.28)]
            img[..., c] += np.sin(xx * fx + yy * fy + ph)
    img = ((img - img.min()) / (img.max() - img.min() + 1e-9) * 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(d, f"{i:05d}.png"))


def throttled():
    try:
        for ln in open("/sys/fs/cgroup/cpu.stat"):
            if ln.startswith("nr_throttled") or ln.startswith("throttled_usec"):
                yield ln.strip()
    except OSError:
        return


if __name__ == "__main__":
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    print("os.cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)))
    for p in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us",
              "/sys/fs/cgroup/cpuset.cpus.effective", "/proc/loadavg"):
        show(p)
    os.system("lscpu | egrep 'Model name|Socket|Thread|Core|MHz|NUMA node\\(s\\)' | head -12")
    d = tempfile.mkdtemp(prefix="tise_decode_")
    t0 = time.perf_counter()
    with ProcessPoolExecutor(64) as ex:
        list(ex.map(write, [(d, i) for i in range(N)], chunksize=64))
    print(f"wrote {N} PNGs in {time.perf_counter() - t0:.1f} s")
    files = sorted(os.path.join(d, f) for f in os.listdir(d))
    t0 = time.perf_counter()
    for f in files[:300]:
        np.asarray(Image.open(f).convert("RGB"))
    print(f"one thread: {(time.perf_counter() - t0) / 300 * 1e3:.2f} ms per image (open + decode + convert)")
    from tise_toolbox_amd import png_ring
    for workers in (8, 16, 32, 64, 128, 32):
        before = list(throttled())
        t0 = time.perf_counter()
        ld = png_ring.PngRingLoader(files, 50, "cpu", workers=workers, chunk=8)
        n, first = 0, None
        for lo, v in ld.iter_host():
            if first is None:
                first = time.perf_counter() - t0
            n += len(v)
        dt = time.perf_counter() - t0
        print(f"ring, {workers:3d} workers: {n} images in {dt:.2f} s, first chunk after {first:.2f} s -> {(n - 8) / (dt - first):7.0f} images/s after it "
              f"({(dt - first) / (n - 8) * workers * 1e3:.2f} ms per image and worker)   cgroup {before} -> {list(throttled())}", flush=True)
