#!/usr/bin/env python3
"""What dist.warm_up_async buys a CLI that runs in a process group: the README recipe (reference .npz + N PNG files) in a ONE-rank RCCL
group forced through the product's own init path (TISE_DIST_FORCE=1: the communicator, the all-reduce of the statistics are
real), weights from a checkpoint FILE (no model broadcast), alternating TISE_DIST_WARMUP=0 (the first all-reduce after the image
loop creates the communicator) and 1 (a helper thread creates it beside the start-up and the image loop).
    python tools/comm_warmup_probe.py [N_IMAGES] [ROUNDS]
Needs tools/probes/comm_warmup.patch applied (dist.warm_up_async / join_warm_up, no broadcast of weights read from a file): the
change was measured (profiles/r06ah_comm_warmup_probe.txt: 0.15-0.35 s of ~0.95 hidden -- the main thread's HIP calls wait for
the communicator's creation) and NOT kept."""
import os
import subprocess
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch  # noqa: E402

from cli_probe import write  # noqa: E402  (the PNG writer of tools/cli_probe.py)

N = int(sys.argv[1]) if len(sys.argv) > 1 else 12000
ROUNDS = int(sys.argv[2]) if len(sys.argv) > 2 else 3

if __name__ == "__main__":
    from concurrent.futures import ProcessPoolExecutor
    from tise_toolbox_amd import png_ring
    from tise_toolbox_amd.inception import build_inception3
    root = tempfile.mkdtemp(prefix="tise_warm_")
    d = os.path.join(root, "gen")
    os.makedirs(d)
    with ProcessPoolExecutor(png_ring.usable_cpus()) as ex:
        list(ex.map(write, [(d, i) for i in range(N)], chunksize=64))
    wfile = os.path.join(root, "inception_standin.pth")
    torch.save(build_inception3(seed=0).state_dict(), wfile)
    ref = os.path.join(root, "ref.npz")
    cli = [sys.executable, "-m", "tise_toolbox_amd.fid_score", "--weights", wfile, "--batch-size", "50"]
    subprocess.run(cli + ["--path2", d, "--save-stats", ref], check=True, capture_output=True)
    for i in range(ROUNDS):
        for warm, force in (("0", "1"), ("1", "1"), ("1", "0")):
            env = dict(os.environ, TISE_TIMING="1", TISE_DIST_FORCE=force, TISE_DIST_WARMUP=warm)
            t0 = time.perf_counter()
            r = subprocess.run(cli + ["--path1", ref, "--path2", d], capture_output=True, text=True, env=env)
            dt = time.perf_counter() - t0
            timing = " | ".join(ln.replace("[tise timing] ", "").replace(" after process start", "") for ln in r.stderr.splitlines() if "[tise timing]" in ln)
            what = "no process group          " if force == "0" else f"one-rank RCCL group, warm-up {'ON ' if warm == '1' else 'OFF'}"
            print(f"{what}: {dt:5.2f} s wall for {N} files   {r.stdout.strip().splitlines()[-1] if r.stdout else r.stderr[-300:]}\n    {timing}", flush=True)
