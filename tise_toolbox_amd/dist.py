"""Data-parallel plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL on ROCm).

The reference evaluators are single-process (no collective anywhere), so this
layer is new: images are independent through decode / resize / trunk, and both reductions are
additive -- FID in {n, sum x, sum x x^T} (fp64, 33.57 MB at d = 2048), IS* in per-split
{A_k, B_kc} (80 KB) -- so the only exchange is ONE all-reduce(SUM) per image set at the end.

Sharding rule: the global file list (os.walk order) is cut to
n_used = (N // batch) * batch (DataLoader(drop_last=True), fid_score.py:215-217), then split
into contiguous index ranges, one per rank; IS* split membership is computed from the GLOBAL
index, so a shard may straddle split borders.
"""
import os

import torch
import torch.distributed as dist


def env_world():
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


_FORCED = False      # a ONE-rank group was brought up on purpose: the collectives below then really run


def _collective():
    """True when the reductions below have to call torch.distributed: more than one rank, or a one-rank group that was
    forced (init_from_env(force=True) / TISE_DIST_FORCE=1: the RCCL communicator, all-reduce and reduce of the job
    execute on a single GPU -- tests/test_gpu_rccl.py, bench.py's `collective` object)."""
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or _FORCED)


def _ipc_default(world, backend):
    """HSA_ENABLE_IPC_MODE_LEGACY=0 for a multi-rank RCCL group, set BEFORE anything initialises HIP in this process: ROCr reads
    its HSA_* switches once, at hsa_init, and torch.cuda.is_available() / set_device already run it.  Measured in
    round 5 (tools/rccl_probe.py, profiles/r05a_rccl_world1.txt): device memory shared between two PROCESSES on this driver
    (hipIpcGetMemHandle / OpenMemHandle -- RCCL's intra-node transport) works with the variable = 0 and never arrives with it
    unset; a one-rank group comes up either way and leaves the environment alone.  An explicit value in the environment wins.
    torchrun users may export it themselves; the package also applies this default at import (tise_toolbox_amd/__init__.py)
    when WORLD_SIZE > 1, which is earlier still."""
    if world > 1 and backend != "gloo":
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")


def collective_timeout():
    """Deadline of every collective of the job's process group (TISE_DIST_TIMEOUT_S, default 120 s): a rank that dies
    mid-loop must not leave the others in barrier() / all_reduce until the launcher's own limit."""
    import datetime
    return datetime.timedelta(seconds=float(os.environ.get("TISE_DIST_TIMEOUT_S", "120")))


def init_from_env(backend=None, force=None):
    """Initialise the default process group from torchrun's environment (no-op for world size 1 unless ``force`` /
    TISE_DIST_FORCE=1 asks for a one-rank group, whose collectives are then executed like any other group's).
    The group carries collective_timeout(); see _ipc_default for HSA_ENABLE_IPC_MODE_LEGACY."""
    global _FORCED
    rank, world, local_rank = env_world()
    _ipc_default(world, backend or os.environ.get("TISE_DIST_BACKEND"))        # first: no torch.cuda call has run in here yet
    if force is None:
        force = os.environ.get("TISE_DIST_FORCE", "0") == "1"
    if (world > 1 or force) and not dist.is_initialized():
        if backend is None:
            # TISE_DIST_BACKEND=gloo lets the multi-rank path be exercised on a single-GPU box (tests only)
            backend = os.environ.get("TISE_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:
            import socket                                   # a forced one-rank group has no launcher: any free port
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist.init_process_group(backend=backend, rank=rank, world_size=world, timeout=collective_timeout())
        _FORCED = world == 1
    return rank, world, local_rank


def run_cli(main):
    """``if __name__ == "__main__"`` body of the CLIs.  One process: ``main()`` and nothing else.  A rank of a multi-process
    job that raises must END at once with a non-zero code (traceback printed): the interpreter's normal shutdown would try
    to tear the process group down and can wait for peers that are themselves waiting in a collective for this rank --
    the launcher (torchrun) takes the other ranks down as soon as one of them has exited non-zero."""
    import sys
    if env_world()[1] <= 1:
        out = main()
        # The result is written and printed: leave at once.  The interpreter's shutdown of a process that holds a HIP context and
        # ~45 GiB of cached device memory took 0.3-0.4 s of the README recipe's 3.2 (round 6, tools/cli_child_probe.py: the last
        # phase stamp 2.8 s after process start, the process gone at 3.2) -- the operating system reclaims all of it anyway.
        # Only on success (an exception takes the normal road with its traceback); TISE_FAST_EXIT=0: the normal shutdown.
        if os.environ.get("TISE_FAST_EXIT", "1") != "0":
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)
        return out
    import traceback
    try:
        return main()
    except SystemExit:
        raise
    except BaseException:                                              # noqa: BLE001
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)


def shutdown():
    """Destroy the default group (a forced one-rank group included)."""
    global _FORCED
    if dist.is_available() and dist.is_initialized():
        dist.destroy_process_group()
    _FORCED = False


def n_used_images(n_images, batch_size):
    """fid_score.py:90-96,215-217: the tail N mod batch_size (in walk order) is dropped."""
    if batch_size <= 0:
        raise ValueError("batch_size must be positive")
    return (n_images // batch_size) * batch_size


def shard_range(n, rank, world):
    """Contiguous [lo, hi) of rank `rank` over n items; sizes differ by at most one."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    hi = lo + base + (1 if rank < rem else 0)
    return lo, hi


def shard_files(files, batch_size, rank, world):
    """The slice of the (os.walk-ordered) file list that rank `rank` evaluates: the drop-last rule is
    applied GLOBALLY (fid_score.py:215-217), then whole batches are dealt out as contiguous ranges."""
    n_used = n_used_images(len(files), batch_size)
    lo, hi = shard_range(n_used // batch_size, rank, world)
    return files[lo * batch_size:hi * batch_size], lo * batch_size


def all_reduce_sum_(t):
    """In-place SUM all-reduce of a tensor (fp64 sufficient statistics); identity for world size 1."""
    if _collective():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def reduce_sum_(t, dst):
    """In-place SUM reduce of ``t`` to rank ``dst`` (other ranks' copies are left undefined); identity for one process.
    RCCL: a real reduce (a ring moves the buffer once instead of twice).  gloo has no reduce for device tensors
    (tests on a single-GPU box, TISE_DIST_BACKEND=gloo): an all-reduce gives the owner the same sum."""
    if _collective():
        if dist.get_backend() == "nccl" or not t.is_cuda:
            dist.reduce(t, dst=dst, op=dist.ReduceOp.SUM)
        else:
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def class_owners(names, world):
    """Per-class O-FID (fid_score.calculate_per_class_fid): class i of the SORTED class list is owned by rank i mod world --
    its statistics are reduced to that rank only, which solves it.  The same map on every rank (the list comes from
    the file names)."""
    return {c: i % world for i, c in enumerate(sorted(names))}


def rank():
    return dist.get_rank() if (dist.is_available() and dist.is_initialized()) else 0


def broadcast_module_(module, src=0):
    """Overwrite every parameter and buffer of `module` with rank `src`'s values (identity for 1 process)."""
    if not (dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1):
        return module
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.numel() == 0:
                continue
            buf = t.detach().clone().contiguous()
            dist.broadcast(buf, src=src)
            t.copy_(buf)
    return module


def world_size():
    """Size of the initialised process group (1 when torch.distributed is not in use)."""
    return dist.get_world_size() if (dist.is_available() and dist.is_initialized()) else 1


def any_rank(flag):
    """True on every rank when `flag` is true on at least one (also a synchronisation point); `flag` itself for one process."""
    if not _collective():
        return bool(flag)
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return bool(t.item() > 0)


def barrier():
    if _collective():
        dist.barrier()


def is_main():
    return (not (dist.is_available() and dist.is_initialized())) or dist.get_rank() == 0
