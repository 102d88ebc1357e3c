#!/usr/bin/env python3
"""Host-inclusive rate of the drop-in CLI: PNG files on disk -> PIL decode in DataLoader workers -> pinned uint8
batches over PCIe -> the device path.  Writes N synthetic 256x256 PNGs, saves reference statistics once, then
times `fid_score` for several worker counts and batch sizes, and with --u8-cache (first run: decode once into the
cache; second run: memory-mapped cache -> pinned double buffer -> side-stream H2D -> device pipeline)."""
import os, sys, time, tempfile, subprocess
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from PIL import Image
from concurrent.futures import ProcessPoolExecutor

N = int(sys.argv[1]) if len(sys.argv) > 1 else 6000


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


if __name__ == "__main__":
    root = tempfile.mkdtemp(prefix="tise_cli_")
    d = os.path.join(root, "gen"); os.makedirs(d)
    t0 = time.perf_counter()
    with ProcessPoolExecutor(min(96, os.cpu_count() or 8)) as ex:
        list(ex.map(write, [(d, i) for i in range(N)], chunksize=64))
    print(f"wrote {N} PNGs in {time.perf_counter() - t0:.1f} s ({sum(os.path.getsize(os.path.join(d, f)) for f in os.listdir(d)) / N / 1e3:.0f} KB each)", flush=True)
    ref = os.path.join(root, "ref.npz")
    env = dict(os.environ)
    base = [sys.executable, "-m", "tise_toolbox_amd.fid_score", "--synthetic-weights", "--path1", ref, "--path2", d,
            "--saved_file", os.path.join(root, "out.txt")]
    subprocess.run([sys.executable, "-m", "tise_toolbox_amd.fid_score", "--synthetic-weights", "--path2", d, "--batch-size", "50",
                    "--save-stats", ref], check=True, capture_output=True, env=env)
    env["TISE_TIMING"] = "1"
    from tise_toolbox_amd import png_ring
    print(f"host: {os.cpu_count()} hardware threads, usable (affinity and cgroup quota) {png_ring.usable_cpus()}; auto workers {png_ring.auto_workers()}", flush=True)
    # round 5: the shared-ring PNG feed (png_ring.py) at 32 / 64 / 128 decode processes and auto, against the DataLoader feed
    quick = len(sys.argv) > 2 and sys.argv[2] == "quick"      # round 6: the README recipe three times, nothing else
    configs = ((50, 0, "ring"),) * 3 if quick else ((50, 8, "ring"), (50, 12, "ring"), (50, 14, "ring"), (50, 16, "ring"), (50, 20, "ring"),
                                                     (50, 32, "ring"), (50, 128, "ring"), (50, 0, "ring"), (50, 0, "ring"),
                                                     (50, 16, "dataloader"), (50, 32, "dataloader"))
    for bs, nw, feed in configs:
        t0 = time.perf_counter()
        r = subprocess.run(base + ["--batch-size", str(bs), "--num-workers", str(nw), "--png-feed", feed], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        feedl = [ln for ln in r.stderr.splitlines() if "png feed" in ln]
        timing = " | ".join(ln.replace("[tise timing] ", "") for ln in r.stderr.splitlines() if "[tise timing]" in ln)
        print(f"{feed:10s} batch {bs:3d} workers {nw:3d}: {dt:6.2f} s wall for {N} images incl. start-up -> {N / dt:7.0f} images/s   "
              f"{r.stdout.strip().splitlines()[-1] if r.stdout else r.stderr[-300:]}\n    {feedl[-1] if feedl else ''}\n    {timing}", flush=True)
    if quick:
        sys.exit(0)
    for bs, label in ((500, "first run (builds the cache, 32 workers)"), (500, "second run (from the cache)"), (500, "third run (from the cache)"),
                      (50, "README batch size, from the cache"), (50, "README batch size, from the cache, again")):
        t0 = time.perf_counter()
        r = subprocess.run(base + ["--batch-size", str(bs), "--num-workers", "32", "--u8-cache"], capture_output=True, text=True, env=env)
        dt = time.perf_counter() - t0
        feed = [ln for ln in r.stderr.splitlines() if "u8 cache feed" in ln]
        print(f"--u8-cache batch {bs}, {label}: {dt:6.1f} s wall incl. start-up -> {N / dt:7.0f} images/s   "
              f"{r.stdout.strip().splitlines()[-1] if r.stdout else r.stderr[-200:]}\n    {feed[-1] if feed else ''}", flush=True)
