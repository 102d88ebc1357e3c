"""CPU, world_size 2, gloo: the data-parallel path -- contiguous index shards, one all-reduce(SUM) of the
fp64 sufficient statistics [S | s | n] and of the IS* split sums, finalisation equal to the single-process
result.  (On the GPU box the same code runs with backend "nccl" = RCCL; the per-shard sums are then
produced by the HIP kernels instead of the numpy emulation used here.)"""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import fid_oracle, is_oracle
    from tests import _cases
    from tise_toolbox_amd import dist as tdist
    r, w, _ = tdist.init_from_env(backend="gloo")
    assert (r, w) == (rank, world) and dist.get_backend() == "gloo"
    d, n_files, bs, C, splits = 48, 1003, 50, 37, 10
    files = [f"{i:05d}.png" for i in range(n_files)]
    shard, base = tdist.shard_files(files, bs, rank, world)
    feats = _cases.pool3_like_features(n_files, d, seed=11)                 # row i = features of file i
    logits = (np.random.default_rng(5).standard_normal((n_files, C)) * 2).astype(np.float32)
    n_used = tdist.n_used_images(n_files, bs)
    idx = np.arange(base, base + len(shard))
    x = feats[idx].astype(np.float64)
    # the device buffer layout [S (d*d) | s (d) | n | pad]
    buf = torch.zeros(d * d + d + 2, dtype=torch.float64)
    buf[:d * d] = torch.from_numpy((x.T @ x).reshape(-1))
    buf[d * d:d * d + d] = torch.from_numpy(x.sum(0))
    buf[d * d + d] = len(idx)
    A, B = is_oracle.is_sums(logits[idx], is_oracle.T_COCO, base, n_used, splits, "coco")
    acc = torch.from_numpy(np.concatenate([A, B.reshape(-1)]))
    tdist.all_reduce_sum_(buf)
    tdist.all_reduce_sum_(acc)
    tdist.barrier()
    if tdist.is_main():
        S = buf[:d * d].numpy().reshape(d, d)
        s = buf[d * d:d * d + d].numpy()
        n = float(buf[d * d + d])
        mu, sigma = fid_oracle.statistics_from_sums(n, s, S)
        mu_ref, sigma_ref = fid_oracle.calculate_activation_statistics(feats[:n_used])
        a = acc.numpy()
        is_m, is_s = is_oracle.is_finalize(a[:splits], a[splits:].reshape(splits, C), n_used, splits, "coco")
        ref_m, ref_s = is_oracle.inception_score_from_logits(logits[:n_used], is_oracle.T_COCO, splits, "coco", dtype=np.float64)
        q.put({"n": n, "n_used": n_used, "mu_err": float(np.abs(mu - mu_ref).max()),
               "sigma_err": float(np.abs(sigma - sigma_ref).max() / np.abs(sigma_ref).max()),
               "is_err": max(abs(is_m - ref_m), abs(is_s - ref_s)), "shard": (base, len(shard))})
    else:
        q.put({"shard": (base, len(shard))})
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_two_rank_allreduce_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    main = [o for o in outs if "n" in o][0]
    assert main["n"] == main["n_used"] == 1000                                # 1003 files, batch 50 -> 3 dropped
    assert sorted(o["shard"] for o in outs) == [(0, 500), (500, 500)]
    assert main["mu_err"] <= 1e-14 and main["sigma_err"] <= 1e-12 and main["is_err"] <= 1e-10


def _worker_perclass(rank, world, port, q):
    """The exchange of fid_score.calculate_per_class_fid on CPU tensors: item shards -> per-class sums -> reduce to the owner
    (tise_toolbox_amd.dist.class_owners / reduce_sum_) -> the owner's scalar -> all-reduce of [value, solved, skipped]."""
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from oracle import fid_oracle
    from tests import _cases
    from tise_toolbox_amd import dist as tdist
    tdist.init_from_env(backend="gloo")
    d = 24
    names = ["cup", "dog", "person", "traffic light", "zebra"]                 # "zebra": a single crop on side 2 -> skipped
    counts = {1: [40, 33, 61, 29, 30], 2: [35, 41, 50, 31, 1]}
    owner = tdist.class_owners(names, world)
    assert owner == {c: i % world for i, c in enumerate(names)} and tdist.rank() == rank
    sums = {}
    for side in (1, 2):
        labels = np.repeat(np.arange(5), counts[side])
        labels = np.random.default_rng(side).permutation(labels)               # classes interleaved in the directory
        feats = _cases.pool3_like_features(len(labels), d, seed=20 + side, shift=0.1 * side).astype(np.float64)
        lo, hi = tdist.shard_range(len(labels), rank, world)                   # this rank's crops
        for i, c in enumerate(names):
            x = feats[lo:hi][labels[lo:hi] == i]
            buf = torch.zeros(d * d + d + 1, dtype=torch.float64)
            buf[:d * d] = torch.from_numpy((x.T @ x).reshape(-1)); buf[d * d:d * d + d] = torch.from_numpy(x.sum(0)); buf[-1] = len(x)
            tdist.reduce_sum_(buf, dst=owner[c])
            sums[side, c] = buf if owner[c] == rank else None
        if rank == 0:
            sums[side, "_all"] = (feats, labels)
    status = torch.zeros((len(names), 3), dtype=torch.float64)
    for i, c in enumerate(names):
        if owner[c] != rank:
            continue
        b1, b2 = sums[1, c], sums[2, c]
        if b1[-1] < 2 or b2[-1] < 2:
            status[i, 2] = 1.0
            continue
        st = [fid_oracle.statistics_from_sums(float(b[-1]), b[d * d:d * d + d].numpy(), b[:d * d].numpy().reshape(d, d)) for b in (b1, b2)]
        status[i, 0] = fid_oracle.calculate_frechet_distance(*st[0], *st[1])
        status[i, 1] = 1.0
    tdist.all_reduce_sum_(status)
    out = {"status": status.numpy().tolist()}
    if rank == 0:
        want = {}
        for i, c in enumerate(names):
            xs = [sums[side, "_all"][0][sums[side, "_all"][1] == i] for side in (1, 2)]
            if min(len(x) for x in xs) >= 2:
                want[c] = float(fid_oracle.calculate_frechet_distance(*fid_oracle.calculate_activation_statistics(xs[0]),
                                                                      *fid_oracle.calculate_activation_statistics(xs[1])))
        out["want"] = want
    q.put(out)
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 3])
def test_per_class_solves_are_sharded_over_ranks(world):
    """class i of the sorted list -> rank i mod W (reduce to the owner, solve there, all-reduce of the
    scalars); 2 and 3 ranks reproduce the one-process per-class distances, and every rank ends with every class."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_perclass, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    outs = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = [o for o in outs if "want" in o][0]["want"]
    names = ["cup", "dog", "person", "traffic light", "zebra"]
    assert set(want) == set(names[:4])
    for o in outs:
        st = np.asarray(o["status"])
        assert st[:, 1].tolist() == [1, 1, 1, 1, 0] and st[:, 2].tolist() == [0, 0, 0, 0, 1]
        for i, c in enumerate(names[:4]):
            assert abs(st[i, 0] - want[c]) <= 1e-9 * max(1.0, abs(want[c])), (c, st[i, 0], want[c])


def test_shard_files_global_drop_last():
    from tise_toolbox_amd import dist as tdist
    files = list(range(1003))
    got = []
    for r in range(8):
        shard, base = tdist.shard_files(files, 50, r, 8)
        assert len(shard) % 50 == 0 and (not shard or shard[0] == base)
        got += shard
    assert got == files[:1000]
    shard, base = tdist.shard_files(list(range(30)), 64, 0, 2)                 # N < batch: nothing is used
    assert shard == [] and base == 0


def test_bench_strong_scaling_shards_cover_the_job_once():
    """bench.py --scaling strong (SURVEY 8(d) Config 3): ONE 30k set, contiguous 30000/N images per rank, device
    batches that divide a rank's range (5000/5000/3750/3750 at 1/2/4/8 GPUs: the largest divisor <= 5000)."""
    sys.path.insert(0, ROOT)
    import bench
    from tise_toolbox_amd import dist as tdist
    want_batch = {1: 5000, 2: 5000, 4: 3750, 8: 3750}
    for world in (1, 2, 3, 4, 8):
        covered = []
        for r in range(world):
            lo, hi = tdist.shard_range(30000, r, world)
            rb = bench.rank_batch(hi - lo)
            if world in want_batch:
                assert rb == want_batch[world] and (hi - lo) % rb == 0
            chunks = [(a, min(a + rb, hi - lo)) for a in range(0, hi - lo, rb)]
            assert chunks[0][0] == 0 and chunks[-1][1] == hi - lo
            covered += [(lo + a, lo + b) for a, b in chunks]
        assert covered[0][0] == 0 and covered[-1][1] == 30000
        assert all(x[1] == y[0] for x, y in zip(covered[:-1], covered[1:]))
    assert bench.rank_batch(1250) == 1250 and bench.rank_batch(100) == 100 and bench.rank_batch(10000) == 5000 and bench.rank_batch(14000) == 3500
    assert bench.rank_batch(10006) == 5000                             # 2 x 5003: no divisor in 128 .. 5000 -> 5000 with a short tail


def test_bench_self_launch_starts_fresh_ranks(monkeypatch):
    """`python bench.py --gpus N` without a torchrun environment starts N ranks itself: the parent
    spawns `python -m torch.distributed.run --nproc-per-node N ... bench.py <same flags>` as a child process (no exec,
    no GPU call in the parent), on 127.0.0.1 with a free port, and returns the child's exit code."""
    sys.path.insert(0, ROOT)
    import subprocess
    import bench
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    assert bench._self_launch(8) == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0" and seen["env"]["TISE_BENCH_LAUNCHER"] == "self"
    # and main() takes that branch before importing anything that touches the GPU
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7


_DEAD_RANK_SCRIPT = r"""
import os, sys, time
sys.path.insert(0, {root!r})
import torch
from tise_toolbox_amd import dist as tdist
rank, world, _ = tdist.init_from_env(backend="gloo")
buf = torch.ones(1 << 16, dtype=torch.float64)
tdist.all_reduce_sum_(buf)                       # the group works
assert float(buf[0]) == world
for step in range(50):                           # "image loop"
    time.sleep(0.02)
    if rank == 1 and step == 10:
        os._exit(9)                              # killed mid-loop: no clean-up, no goodbye
tdist.all_reduce_sum_(buf)                       # rank 0 arrives alone
tdist.barrier()
print("rank", rank, "finished", flush=True)
"""


@pytest.mark.timeout(200)
def test_a_rank_that_dies_mid_loop_takes_the_job_down(tmp_path):
    """world-2 gloo job, rank 1 is killed in the middle of its loop.  Rank 0 must not sit in the all-reduce
    until a launcher gives up: the group carries dist.collective_timeout() (TISE_DIST_TIMEOUT_S, default 120 s) and the lost
    peer surfaces as an exception -> rank 0 exits non-zero, well inside 150 s."""
    import subprocess
    import time
    script = tmp_path / "dead_rank.py"
    script.write_text(_DEAD_RANK_SCRIPT.format(root=ROOT))
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   TISE_DIST_BACKEND="gloo", TISE_DIST_TIMEOUT_S="30")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    t0 = time.time()
    out1, err1 = procs[1].communicate(timeout=150)
    out0, err0 = procs[0].communicate(timeout=150)
    took = time.time() - t0
    assert procs[1].returncode == 9
    assert procs[0].returncode not in (0, None), (out0, err0[-500:])
    assert "finished" not in out0 and took < 150, took
