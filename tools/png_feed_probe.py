"""Timeline of the from-PNG-files job (bench.py's png_feed leg) for several decode-process counts: when the workers are up, when
the first / every device batch is handed over, when the last file is decoded, when the job ends -- against the resident rate.
    python tools/png_feed_probe.py [N_IMAGES] [WORKERS,WORKERS,...]"""
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from tise_toolbox_amd import png_ring  # noqa: E402
from tise_toolbox_amd.engine import RealismEngine, T_COCO, device_batch_images, frechet_solver  # noqa: E402


_thr = [0]


def thr():
    """CFS throttling of this cgroup since the last call: periods throttled / microseconds."""
    try:
        d = dict(l.split() for l in open("/sys/fs/cgroup/cpu.stat"))
        now = (int(d["nr_throttled"]), int(d["throttled_usec"]))
    except Exception:
        return "n/a"
    last = _thr[0] or now
    _thr[0] = now
    return f"{now[0] - last[0]} periods / {(now[1] - last[1]) / 1e3:.0f} ms"


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
    counts = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [16, 14, 12, 8]
    dev = torch.device("cuda", 0)
    eng = RealismEngine(dims=2048, device_index=0, seed=0, with_logits=True)
    data = torch.empty((n, 256, 256, 3), dtype=torch.uint8, device=dev)
    for i in range(0, n, 1000):
        data[i:i + 1000] = bench.synth_images_device(i, min(i + 1000, n), dev, seed=0)
    tmp = tempfile.mkdtemp(prefix="tise_feedprobe_")
    npy = os.path.join(tmp, "pixels.npy")
    np.save(npy, data.cpu().numpy())
    d = os.path.join(tmp, "png")
    os.makedirs(d)
    from concurrent.futures import ProcessPoolExecutor
    procs = png_ring.usable_cpus()
    step = -(-n // (4 * procs))
    with ProcessPoolExecutor(procs) as ex:
        list(ex.map(bench._write_pngs, [(npy, a, min(a + step, n), d) for a in range(0, n, step)]))
    files = [os.path.join(d, f"{i:06d}.png") for i in range(n)]
    limit = device_batch_images(50)
    solver = frechet_solver(2048, dev)
    eng.begin(n_total=n)
    for a in range(0, n, limit):
        eng.step_u8(data[a:a + limit], a)
    mu_ref, sigma_ref = eng.statistics()

    def tail():
        eng.reduce()
        mu, sigma = eng.statistics()
        solver.distance(mu, sigma, mu_ref, sigma_ref)
        eng.inception_score()
        torch.cuda.synchronize()

    def resident(sizes):
        eng.begin(n_total=n, temperature=T_COCO, splits=10, rule="coco")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        a = 0
        for r in sizes:
            eng.step_u8(data[a:a + r], a)
            a += r
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        tail()
        return t1 - t0, time.perf_counter() - t0

    for _ in range(2):
        loop, wall = resident([limit] * (n // limit))
    print(f"resident, batches of {limit}: loop {loop * 1e3:.1f} ms, job {wall * 1e3:.1f} ms ({n / wall:.0f} img/s)")
    for w in counts:
        for rep in range(2):
            torch.cuda.synchronize()
            thr()
            t0 = time.perf_counter()
            loader = png_ring.PngRingLoader(files, 50, dev, group=limit // 50, workers=w)
            t_spawn = time.perf_counter() - t0
            eng.begin(n_total=n, temperature=T_COCO, splits=10, rule="coco")
            marks, base = [], 0
            for big in loader:
                marks.append((time.perf_counter() - t0, big.shape[0]))
                eng.step_u8(big, base)
                base += big.shape[0]
            t_loop = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_sync = time.perf_counter() - t0
            tail()
            wall = time.perf_counter() - t0
        sizes = [m[1] for m in marks]
        rl, rw = resident(sizes)
        print(f"workers {w:3d}: spawn {t_spawn * 1e3:5.1f} ms | handed at " + " ".join(f"{m[0] * 1e3:.0f}" for m in marks)
              + f" ms | all decoded {loader.decode_seconds * 1e3:.0f} ms | loop returned {t_loop * 1e3:.0f}, device idle {t_sync * 1e3:.0f}, job {wall * 1e3:.0f} ms "
              f"= {n / wall:.0f} img/s | feeder waited {loader.wait_decode_seconds * 1e3:.0f} ms for decode, {loader.wait_buffer_seconds * 1e3:.0f} ms for a buffer, {loader.enqueue_seconds * 1e3:.0f} ms in memcpy calls, {loader.wait_copy_seconds * 1e3:.0f} ms for copies; throttled {thr()} | resident with the same batches {rl * 1e3:.0f} / {rw * 1e3:.0f} ms")
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
