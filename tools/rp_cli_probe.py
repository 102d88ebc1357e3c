#!/usr/bin/env python3
"""Host-inclusive time of the RP-COCO drop-in CLI (BASELINE configs[3] shape on one GPU): N items, 100 candidate captions
each from a pool of 40 k distinct captions, one 256 x 256 PNG per item.  Writes the PNGs and the pickle, then runs
`python -m tise_toolbox_amd.RP_coco` with the ring feed (decode processes -> pinned ring -> clip's preprocess on the device)
and with the DataLoader feed (Pillow preprocess on worker processes), whole process wall time each."""
import os, pickle, subprocess, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from concurrent.futures import ProcessPoolExecutor
from tools.host_decode_probe import write

N = int(sys.argv[1]) if len(sys.argv) > 1 else 30000


def wr(args):
    d, i = args
    write((d, i))
    os.rename(os.path.join(d, f"{i:05d}.png"), os.path.join(d, f"{i}.png"))


if __name__ == "__main__":
    root = tempfile.mkdtemp(prefix="tise_rp_")
    d = os.path.join(root, "img"); os.makedirs(d)
    t0 = time.perf_counter()
    with ProcessPoolExecutor(16) as ex:
        list(ex.map(wr, [(d, i) for i in range(N)], chunksize=64))
    rng = np.random.default_rng(0)
    words = ["red", "blue", "cat", "dog", "table", "sky", "tree", "car", "two", "three", "on", "under", "near", "a", "the", "small", "large", "bird", "boat", "cup"]
    pool = [" ".join(rng.choice(words, size=int(rng.integers(6, 14)))) + f" {k}" for k in range(40000)]
    items = [{"caption_id": i, "caption": pool[i % 40000], "mismatched_captions": [pool[int(j)] for j in rng.integers(0, 40000, 99)]} for i in range(N)]
    pk = os.path.join(root, "rp.pkl")
    pickle.dump(items, open(pk, "wb"))
    print(f"wrote {N} PNGs + pickle in {time.perf_counter() - t0:.1f} s", flush=True)
    base = [sys.executable, "-m", "tise_toolbox_amd.RP_coco", "--image_dir", d, "--rp_input_file", pk, "--saved_file_path", os.path.join(root, "rp.txt"),
            "--synthetic-weights", "--seed", "0"]
    for feed in ("ring", "dataloader", "ring"):
        t0 = time.perf_counter()
        r = subprocess.run(base + ["--png-feed", feed], capture_output=True, text=True, env=dict(os.environ, TISE_TIMING="1"))
        dt = time.perf_counter() - t0
        last = r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]
        ph = " | ".join(ln.replace("[tise timing] ", "") for ln in r.stderr.splitlines() if "[tise timing]" in ln)
        print(f"--png-feed {feed:10s}: {dt:6.2f} s wall for {N} items incl. start-up -> {N / dt:7.0f} items/s   {last}\n    {ph}", flush=True)
