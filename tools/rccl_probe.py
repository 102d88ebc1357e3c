#!/usr/bin/env python3
"""RCCL on ONE GPU: the collective leg of the job (BASELINE configs[2]) executed through the product's own code path.

No 8-GPU node is available to this build, so this is the proxy one GPU allows.  Each leg runs in a
FRESH child process (this parent never touches the GPU and never execs):

  world1   WORLD_SIZE=1, backend "nccl" forced through tise_toolbox_amd.dist.init_from_env(force=True): communicator
           init, all-reduce of the real 33.57 MB [S | s | n] statistics buffer (tise_stats_buffer) and of the 80 KB IS*
           sums through RealismEngine.reduce(), dist.reduce to rank 0 (the per-class owner exchange), any_rank, barrier,
           destroy.  Values must be unchanged (a one-rank sum is the identity) and finalize must still give np.cov.
  ipc      what HSA_ENABLE_IPC_MODE_LEGACY governs: hipIpcGetMemHandle / hipIpcOpenMemHandle between two processes
           (torch.multiprocessing shares a CUDA tensor with a spawned child on the same GPU -- the mechanism RCCL's
           intra-node P2P transport uses between ranks).

Both legs run twice: with HSA_ENABLE_IPC_MODE_LEGACY=0 and with the variable REMOVED from the child's environment.
Prints one JSON object; `python tools/rccl_probe.py > profiles/r05x_rccl_world1.txt`.
"""
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def leg_world1():
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    import torch.distributed as td
    from tise_toolbox_amd import device, dist as tdist
    t0 = time.perf_counter()
    rank, world, local_rank = tdist.init_from_env(force=True)
    t_init = time.perf_counter() - t0
    assert td.is_initialized() and td.get_world_size() == 1
    dev = torch.device("cuda", local_rank)
    out = {"backend": td.get_backend(), "world_size": world, "init_process_group_s": t_init,
           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    d = 2048
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.rand((3000, d), generator=g).to(dev)
    acc = device.StatsAccumulator(d, dev)
    acc.update(x)
    buf = acc.buffer()
    out["stats_buffer_bytes"] = int(buf.numel() * 8)
    before = buf.clone()
    isacc = device.InceptionScoreAccumulator(1000, 3000, 0.9091363549232483, 10, "coco", False, dev)
    isacc.update(torch.randn((3000, 1000), generator=g).to(dev), 0)
    is_before = isacc.acc.clone()

    def timed(fn, reps):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    t0 = time.perf_counter()
    tdist.all_reduce_sum_(buf)                      # first call: lazy communicator creation (ncclCommInitRank)
    torch.cuda.synchronize()
    out["first_allreduce_incl_comm_init_s"] = time.perf_counter() - t0
    out["allreduce_stats_ms"] = timed(lambda: tdist.all_reduce_sum_(buf), 20)
    out["allreduce_is_sums_ms"] = timed(lambda: tdist.all_reduce_sum_(isacc.acc), 20)
    out["reduce_to_owner_ms"] = timed(lambda: tdist.reduce_sum_(buf, 0), 20)
    assert torch.equal(buf, before) and torch.equal(isacc.acc, is_before), "a one-rank SUM must be the identity"
    assert tdist.any_rank(True) is True and tdist.any_rank(False) is False
    t0 = time.perf_counter()
    tdist.barrier()
    out["barrier_ms"] = (time.perf_counter() - t0) * 1e3
    mu, sigma = acc.finalize()
    xs = x.double().cpu().numpy()
    out["sigma_err_vs_npcov"] = float(np.abs(sigma.cpu().numpy() - np.cov(xs, rowvar=False)).max())
    assert out["sigma_err_vs_npcov"] <= 1e-12
    # the engine's own reduce() on a real (small) image set
    from tise_toolbox_amd.engine import RealismEngine
    eng = RealismEngine(dims=2048, device_index=local_rank, seed=0, with_logits=True)
    eng.begin(n_total=16)
    imgs = (torch.rand((16, 256, 256, 3), generator=g) * 255).to(torch.uint8).to(dev)
    eng.step_u8(imgs, 0)
    s0 = eng.stats.buffer().clone()
    eng.reduce()
    torch.cuda.synchronize()
    assert torch.equal(eng.stats.buffer(), s0)
    out["engine_reduce_ok"] = True
    t0 = time.perf_counter()
    tdist.shutdown()
    out["destroy_s"] = time.perf_counter() - t0
    assert not td.is_initialized()
    out["rccl_version"] = ".".join(str(v) for v in torch.cuda.nccl.version()) if hasattr(torch.cuda, "nccl") else None
    print("LEG " + json.dumps(out), flush=True)


def _ipc_child(q, done):
    import torch
    t = q.get()
    ok = bool((t == 7).all().item())
    t += 1
    torch.cuda.synchronize()
    done.put(ok)


def leg_ipc():
    import torch
    import torch.multiprocessing as mp
    out = {"HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")}
    ctx = mp.get_context("spawn")
    q, done = ctx.Queue(), ctx.Queue()
    p = ctx.Process(target=_ipc_child, args=(q, done))
    p.start()
    try:
        t = torch.full((1 << 20,), 7, dtype=torch.int32, device="cuda:0")
        torch.cuda.synchronize()
        q.put(t)                                   # hipIpcGetMemHandle here, hipIpcOpenMemHandle in the child
        ok = done.get(timeout=float(os.environ.get("TISE_IPC_PROBE_TIMEOUT", "120")))
        p.join(60)
        torch.cuda.synchronize()
        out["child_read_ok"] = ok
        out["child_write_visible"] = bool((t == 8).all().item())
        out["ok"] = ok and out["child_write_visible"]
    except Exception as e:                          # noqa: BLE001
        out["ok"] = False
        out["error"] = f"{type(e).__name__}: {e}"[:600]
        if p.is_alive():
            p.kill()
    print("LEG " + json.dumps(out), flush=True)


def run_child(leg, with_var):
    env = dict(os.environ)
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    if with_var:
        env["HSA_ENABLE_IPC_MODE_LEGACY"] = "0"
    env.update({"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(_free_port())})
    env.pop("TISE_DIST_BACKEND", None)
    t0 = time.perf_counter()
    try:
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--leg", leg], env=env, capture_output=True, text=True, timeout=600)
        rc, so, se = r.returncode, r.stdout, r.stderr
    except subprocess.TimeoutExpired as e:
        rc, so, se = -9, (e.stdout or b"").decode() if isinstance(e.stdout, bytes) else (e.stdout or ""), "timeout"
    res = {"returncode": rc, "wall_s": time.perf_counter() - t0}
    for line in so.splitlines():
        if line.startswith("LEG "):
            res.update(json.loads(line[4:]))
    if rc != 0:
        res["stderr_tail"] = se[-1500:]
    return res


def main():
    if "--leg" in sys.argv:
        leg = sys.argv[sys.argv.index("--leg") + 1]
        {"world1": leg_world1, "ipc": leg_ipc}[leg]()
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    legs = args or ["world1", "ipc"]
    out = {}
    for leg in legs:
        out[leg] = {"with HSA_ENABLE_IPC_MODE_LEGACY=0": run_child(leg, True)}
        if "--with-var-only" not in sys.argv:
            out[leg]["variable unset"] = run_child(leg, False)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
