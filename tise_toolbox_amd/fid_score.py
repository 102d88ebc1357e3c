#!/usr/bin/env python3
"""FID on MI355X -- drop-in for the reference ``image_realism/FID/fid_score.py``.

Same module-level functions (names, argument meaning, return types, error behaviour) and the
same CLI flags / result-file text as the reference, so a caller switches by changing the import:

    get_activations                 fid_score.py:67      calculate_activation_statistics  :174
    calculate_frechet_distance      :121                 _compute_statistics_of_path      :199
    calculate_fid_given_paths       :223                 CLI                              :51-64,:241-254

What runs where: PNG decode on DataLoader workers (as the reference, :215-217); resize + input
affine, InceptionV3, fp64 mean/covariance accumulation and the Frechet distance on the GPU
(``engine.py`` -> ``libtise_hip.so`` + PyTorch-ROCm).  With ``torchrun --nproc-per-node N`` the file
list is sharded over N GPUs and the sufficient statistics are combined with one RCCL all-reduce.

Deviations from the reference, all deliberate (SURVEY.md notes N3-N5):
  * ``--gpu ""`` (CPU mode) is refused: this package is the MI355X path and has no CPU fallback.
  * ``--path1/--path2`` are required (upstream defaults them to the int 64 by a copy-paste slip).
  * extra flags: ``--weights`` (torchvision-format state_dict; there is no network to download it; without the flag
    the file the reference would read is looked up, see ``weights.py``), ``--synthetic-weights`` + ``--seed``
    (seeded stand-in parameters for plumbing / throughput runs: results are tagged), ``--save-stats`` (write
    mu/sigma .npz, the format upstream only reads; with ``--path1`` omitted this is the STATS-ONLY mode: no
    Frechet distance is solved), ``--label`` ("FID" | "O-FID", object_fidelity/O-FID/fid_score.py:216-222),
    ``--num-classes`` (80 for the O-FID fine-tune, O-FID/inception.py:58-64), ``--per-class`` (extension, BASELINE
    configs[4]: one FID per object class, crops grouped by the ``{class}`` token of ``{stem}_{class}_{k}.png``,
    object_fidelity/crop_object.py:45).
"""
import contextlib
import os
import sys
import time
_T_IMPORT0 = time.time()
import warnings
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import numpy as np
import torch
import torch.utils.data

from . import _lib, device, dist as tdist, img_data, weights as tweights
from .engine import (RealismEngine, coalesce_batches, coalesce_u8, device_batch_images, frechet_solver, require_gpu,
                     run_with_exact_fallback)
from .inception import InceptionV3

warnings.filterwarnings("ignore")          # fid_score.py:49
_T_IMPORT1 = time.time()


def _timing(label, t0=None):
    """TISE_TIMING=1: phase times of the process on stderr (tools/cli_probe.py reads them)."""
    if os.environ.get("TISE_TIMING") == "1" and tdist.is_main():
        now = time.time()
        try:
            # age of the process from the kernel's own clocks: start time in ticks since boot (/proc/self/stat field 22) against
            # /proc/uptime.  (psutil's create_time adds the start ticks to /proc/stat's btime, a WHOLE second: its phase stamps
            # were off by up to 1 s and could exceed the wall clock bench.py's cli_process leg measured around the process)
            start_ticks = int(open("/proc/self/stat").read().rsplit(")", 1)[1].split()[19])
            born = now - (float(open("/proc/uptime").read().split()[0]) - start_ticks / os.sysconf("SC_CLK_TCK"))
        except Exception:                                              # noqa: BLE001
            born = _T_IMPORT0
        extra = f" (+{now - t0:.2f} s)" if t0 is not None else ""
        print(f"[tise timing] {label}: {now - born:.2f} s after process start{extra}", file=sys.stderr, flush=True)
    return time.time()


def _build_parser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter)
    parser.add_argument("--batch-size", type=int, default=64, help="Batch size to use")
    parser.add_argument("--dims", type=int, default=2048, choices=list(InceptionV3.BLOCK_INDEX_BY_DIM),
                        help=("Dimensionality of Inception features to use. " "By default, uses pool3 features"))
    parser.add_argument("-c", "--gpu", default="0", type=str, help="GPU to use (CPU mode is not provided)")
    parser.add_argument("--path1", type=str, default=None, help="omit together with --save-stats: statistics of "
                        "--path2 only, no Frechet distance")
    parser.add_argument("--path2", type=str, required=True)
    parser.add_argument("--saved_file", type=str, default="")
    parser.add_argument("--weights", type=str, default=None, help="torchvision-format InceptionV3 state_dict (.pth)")
    parser.add_argument("--num-classes", type=int, default=1000)
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="seeded stand-in parameters (plumbing / throughput only; results are tagged)")
    parser.add_argument("--seed", type=int, default=0, help="seed of the --synthetic-weights parameters")
    parser.add_argument("--u8-cache", action="store_true",
                        help="decode each image directory once into <dir>/.tise_u8_cache.npy and feed later runs (and "
                             "all ranks) from it over a double-buffered pinned host->device pipeline")
    parser.add_argument("--per-class", action="store_true",
                        help="O-FID extension: one FID per object class ({stem}_{class}_{k}.png crops)")
    parser.add_argument("--save-stats", type=str, default="", help="write mu/sigma of --path2 to this .npz")
    parser.add_argument("--label", type=str, default="FID", choices=["FID", "O-FID"])
    parser.add_argument("--num-workers", type=int, default=0,
                        help="PNG decode processes (the reference hard-codes 8 DataLoader workers, fid_score.py:216); 0 = auto: "
                             "min(128, cpus / 2) shared by the ranks of the node")
    parser.add_argument("--conv", type=str, default=None, choices=["split", "exact"],
                        help="split: hand-written split-fp16 MFMA convolutions (fp32-class arithmetic, default); exact: fp32 "
                             "convolutions (MIOpen).  A split run whose range guard fires is finished on the exact path automatically")
    parser.add_argument("--png-feed", type=str, default="ring", choices=["ring", "dataloader"],
                        help="ring: decode workers write into one shared page-locked ring the parent copies from (png_ring.py); "
                             "dataloader: torch DataLoader workers + collate + pin_memory (round 1-4 path, also the fallback "
                             "for directories whose images differ in size)")
    return parser


def _engine_for(model, dims):
    """Wrap a user-supplied reference-style model, or build the default one."""
    if isinstance(model, RealismEngine):
        return model
    eng = getattr(model, "_tise_engine", None)
    if eng is None:
        eng = RealismEngine(dims=dims, model=model, fold_bn=isinstance(model, InceptionV3))
        try:
            model._tise_engine = eng
        except Exception:
            pass
    return eng


def _forward_batch(engine, model, batch):
    """One batch -> (B, dims) fp32 features on the device, for either input convention."""
    if isinstance(batch, (list, tuple)):                 # ragged uint8 crops: resize each into one batch, ONE trunk pass
        return engine.features_from_u8_list(batch)[0]
    if batch.dtype == torch.uint8:                       # (B,H,W,3) decoded images: fused device resize
        return engine.features_from_u8(batch.to(engine.device, non_blocking=True))[0]
    if isinstance(model, InceptionV3):
        return engine.features_from_float(batch)[0]
    # arbitrary nn.Module following the reference contract model(batch)[0] -> (B, dims, h, w)
    with torch.no_grad():
        pred = model(batch.to(engine.device))[0]
        if pred.shape[2] != 1 or pred.shape[3] != 1:     # fid_score.py:110-111
            pred = torch.nn.functional.adaptive_avg_pool2d(pred, output_size=(1, 1))
        return pred.reshape(pred.shape[0], -1).float().contiguous()


def _first(images, n):
    """The first n batches of a loader (fid_score.py:99 iterates all of them; n = its own batch count)."""
    import itertools
    return itertools.islice(iter(images), n)


def _check_cuda(cuda):
    if not cuda:
        raise _lib.TiseLibraryError(
            "cuda=False / --gpu '' requests the reference's CPU path; tise_toolbox_amd is MI355X-only "
            "and has no CPU fallback")
    require_gpu()


def get_activations(images, model, batch_size=64, dims=2048, cuda=True, verbose=True):
    """Activations of the pool_3 layer for all images (fid_score.py:67-118).

    ``images``: sized iterable of batches (``len`` = number of batches, :90) of either
    float (B,3,H,W) tensors in [0,1] (reference convention) or uint8 (B,H,W,3) tensors.
    Returns a float64 numpy array (n_used, dims) -- one device->host copy at the end instead of
    one per batch (:113).
    """
    _check_cuda(cuda)
    model.eval()                                          # :86
    d0 = images.__len__() * batch_size                    # :90
    if batch_size > d0:                                   # :91-93
        print(("Warning: batch size is bigger than the data size. " "Setting batch size to data size"))
        batch_size = d0
    n_batches = d0 // batch_size                          # :95  (ZeroDivisionError for an empty loader, as upstream)
    n_used_imgs = n_batches * batch_size                  # :96
    engine = _engine_for(model, dims)
    pred_dev = torch.empty((n_used_imgs, dims), dtype=torch.float32, device=engine.device)
    start = 0
    # the loader's batches are gathered into device batches (engine.device_batch_images): ``batch_size`` defines the
    # bookkeeping above, not the size of a trunk pass
    limit = device_batch_images(batch_size)
    engine.reserve_activations(min(n_used_imgs, limit))
    for batch in coalesce_u8(_first(images, n_batches), engine.device, limit, (n_used_imgs, batch_size)):   # :99
        f = _forward_batch(engine, model, batch)
        pred_dev[start:start + f.shape[0]] = f.reshape(f.shape[0], -1)                       # :113
        start += f.shape[0]
    if verbose:
        print(" done")                                    # :116
    engine.check_numerics(collective=False)               # split-fp16 range guard (no silent inf / NaN features)
    return pred_dev.cpu().numpy().astype(np.float64)      # :98 pred_arr is float64


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """Frechet distance d^2 = ||mu_1 - mu_2||^2 + Tr(C_1 + C_2 - 2*sqrt(C_1*C_2))  (fid_score.py:121-171).

    Accepts numpy arrays or CUDA tensors; evaluated on the device in fp64 (csrc/frechet.hip).
    Returns np.float64 like the reference.  The reference's rescue branch for a singular
    product (:156-160) is kept: when the device reports non-finite values the message is printed
    and the computation repeated with ``eps`` added to both diagonals.  The complex-residue
    ``ValueError`` (:163-167) cannot occur here: eigenvalues of a symmetric matrix are real.
    """
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())

    def prep(x, nd):
        if isinstance(x, torch.Tensor):
            x = x.to(dev, torch.float64)
            return torch.atleast_1d(x) if nd == 1 else torch.atleast_2d(x)
        x = np.atleast_1d(x) if nd == 1 else np.atleast_2d(x)             # :143-147
        return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float64), device=dev)

    mu1, mu2 = prep(mu1, 1), prep(mu2, 1)
    sigma1, sigma2 = prep(sigma1, 2), prep(sigma2, 2)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"            # :149
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"    # :150
    d = mu1.shape[0]
    solver = frechet_solver(d, dev)
    res = solver.distance(mu1, sigma1, mu2, sigma2, 0.0)
    if res["flags"] & _lib.TISE_FLAG_NONFINITE:                                                        # :156-160
        msg = ("fid calculation produces singular product; " "adding %s to diagonal of cov estimates") % eps
        print(msg)
        res = solver.distance(mu1, sigma1, mu2, sigma2, float(eps))
    calculate_frechet_distance.last_result = res
    return np.float64(res["fid"])


def calculate_activation_statistics(images, model, batch_size=64, dims=2048, cuda=True, verbose=True,
                                    return_device=False):
    """mu = mean(act), sigma = cov(act) of the pool_3 activations (fid_score.py:174-196).

    Unlike the reference no (N, dims) float64 array is ever materialised: every batch is folded
    into fp64 {n, sum x, sum x x^T} on the device and (mu, sigma) are finalised there.  Under
    torchrun each rank passes ITS shard of the batches; the sums are all-reduced over RCCL.
    Returns numpy float64 arrays (or CUDA tensors with ``return_device``).
    """
    _check_cuda(cuda)
    model.eval()
    if tdist.world_size() > 1:
        # data-parallel: `images` is THIS rank's shard and may be empty (fewer batches than ranks).  The reference's
        # single-process bookkeeping below would divide by zero there while the other ranks wait in the all-reduce,
        # so an empty rank contributes zeros and the emptiness check is made once, on the global count.
        n_batches = images.__len__()
    else:
        d0 = images.__len__() * batch_size                # fid_score.py:90-96 bookkeeping
        if batch_size > d0:
            print(("Warning: batch size is bigger than the data size. " "Setting batch size to data size"))
            batch_size = d0
        n_batches = d0 // batch_size                      # ZeroDivisionError for an empty loader, as upstream
    engine = _engine_for(model, dims)
    stats = device.StatsAccumulator(dims, engine.device)
    engine.reserve_activations(min(n_batches * batch_size, device_batch_images(batch_size)))      # one allocation of the passes' peak
    if getattr(images, "pregrouped", False):              # img_data.U8CacheLoader(group=K): items are device batches already
        batches = iter(images)
    else:
        limit = device_batch_images(batch_size)
        batches = coalesce_u8(_first(images, n_batches), engine.device, limit, (n_batches * batch_size, batch_size))
    err = None
    try:
        for batch in batches:
            stats.update(_forward_batch(engine, model, batch))
    except Exception as e:                                # noqa: BLE001 -- re-raised below, on EVERY rank
        err = e
    # A rank whose image loop failed (an unreadable file, a directory of ragged sizes, a dead decode worker) must not leave
    # the others waiting in the all-reduce below: the failure is agreed on collectively and every rank raises.
    if tdist.world_size() > 1:
        if tdist.any_rank(err is not None):
            if err is not None:
                raise err
            raise RuntimeError("the image loop failed on another rank (its own message says why)")
    elif err is not None:
        raise err
    engine.check_numerics()                               # split-fp16 range guard (no silent inf / NaN features)
    tdist.all_reduce_sum_(stats.buffer())
    if tdist.world_size() > 1 and stats.count() == 0:     # identical on every rank (read after the all-reduce)
        raise ZeroDivisionError("no complete batch in the image set (global N < batch size)")
    mu, sigma = stats.finalize()
    if verbose:
        print(" done")
    if return_device:
        return mu, sigma
    return mu.cpu().numpy(), sigma.cpu().numpy()


U8_CACHE_NAME = ".tise_u8_cache.npy"


_PNG_FEED = {"mode": "ring"}        # --png-feed
_RING_PREFETCH = {}                  # directory -> PngRingLoader whose workers are already decoding (started before the model was built)
_RING_LOCK = __import__("threading").Lock()   # the second directory's prefetch is started from the first loader's feeder thread


def _num_workers(num_workers, world=1):
    from . import png_ring
    return int(num_workers) if num_workers and int(num_workers) > 0 else png_ring.auto_workers(world)


def _shard_of_path(path, batch_size):
    files = img_data.get_filenames(path)                  # os.walk order (img_data.py:27-35)
    rank, world, _ = tdist.env_world()
    shard, _ = tdist.shard_files(files, batch_size, rank, world)       # drop_last=True (:215-217), whole batches
    return shard, world


def prefetch_png_ring(path, batch_size, num_workers=0):
    """Start the decode workers of an image directory NOW (before the model is built / while the other side is still in
    the network): _compute_statistics_of_path picks the running loader up.  No GPU call is made here."""
    from . import png_ring
    if _PNG_FEED["mode"] != "ring" or path.endswith(".npz") or not os.path.isdir(path):
        return None
    with _RING_LOCK:
        if path in _RING_PREFETCH:
            return _RING_PREFETCH[path]
        shard, world = _shard_of_path(path, batch_size)
        if not shard:
            return None
        group = device_batch_images(batch_size) // batch_size
        dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_initialized() else torch.device("cuda", tdist.env_world()[2])
        loader = png_ring.PngRingLoader(shard, batch_size, dev, group=group, workers=_num_workers(num_workers, world), start=True)
        _RING_PREFETCH[path] = loader
        return loader


def _compute_statistics_of_path(path, model, batch_size, dims, cuda, num_workers=8, u8_cache=False):
    """fid_score.py:199-220: an .npz holds (mu, sigma); a directory is walked and pushed through the net.
    ``u8_cache``: decode the directory ONCE into ``<path>/.tise_u8_cache.npy`` ((N,H,W,3) uint8, walk order) and feed
    later runs -- and every rank of a data-parallel run -- from it through a double-buffered pinned host->device
    pipeline (img_data.U8CacheLoader) instead of PNG-decoding DataLoader workers."""
    if path.endswith(".npz"):
        f = np.load(path, allow_pickle=True)              # :201-203
        m, s = f["mu"][:], f["sigma"][:]
        f.close()
        return m, s
    files = img_data.get_filenames(path)                  # os.walk order (img_data.py:27-35)
    rank, world, _ = tdist.env_world()
    if u8_cache:
        cache = os.path.join(path, U8_CACHE_NAME)          # the name holds neither "png" nor "jpg": never walked
        # valid only for exactly these files as they are on disk now (relative path, size, mtime_ns of every file in
        # walk order, img_data.files_fingerprint): images regenerated under the same names rebuild it
        err = None
        if tdist.is_main() and not img_data.u8_cache_is_current(cache, files, path):
            try:
                img_data.build_u8_cache(files, cache, _num_workers(num_workers, world), root=path)
            except Exception as e:                         # the other ranks must not wait in a barrier for a cache that never comes
                err = e
        failed = tdist.any_rank(err is not None)
        if err is not None:
            raise err
        if failed:
            raise RuntimeError(f"--u8-cache: rank 0 could not build {cache}")
        n_used = tdist.n_used_images(len(files), batch_size)
        lo, hi = tdist.shard_range(n_used // batch_size, rank, world)
        engine = _engine_for(model, dims)
        shape = np.load(cache, mmap_mode="r").shape
        group = device_batch_images(batch_size, int(np.prod(shape[1:]))) // batch_size
        loader = img_data.U8CacheLoader(cache, batch_size, engine.device, rows=(lo * batch_size, hi * batch_size), group=group)
        t0 = time.perf_counter()
        out = calculate_activation_statistics(loader, model, batch_size, dims, cuda)
        wall = time.perf_counter() - t0
        n_img = len(loader) * batch_size
        if tdist.is_main() and n_img:
            steady = ""
            sec = loader.steady_seconds()
            if sec and n_img > loader.first_item_rows:
                # without the first device batch (first-use costs of the process: code objects, allocator, resize plans):
                # device events at the end of the first and of the last device batch's work (the host time at
                # which the consumer came back for the second item still contained most of the first batch's device time)
                rest = n_img - loader.first_item_rows
                steady = f"; after the first device batch {rest / sec:.0f} images/s"
            print(f"[tise] u8 cache feed: {n_img} images in {wall:.2f} s ({n_img / wall:.0f} images/s on this rank{steady}); host side "
                  f"(page cache -> pinned buffer -> H2D enqueue) {loader.h2d_seconds:.2f} s, the rest is the device pipeline",
                  file=sys.stderr)
        return out
    shard, _ = tdist.shard_files(files, batch_size, rank, world)       # drop_last=True (:215-217), whole batches
    num_workers = _num_workers(num_workers, world)
    if _PNG_FEED["mode"] == "ring":
        # decode workers -> shared page-locked ring -> side-stream H2D (png_ring.py).  A directory with images of different
        # sizes: one process falls back to the DataLoader path below; under torchrun the rank that meets the odd file raises
        # RaggedImages inside calculate_activation_statistics, which agrees on the failure collectively -- every rank raises
        # (the others a generic RuntimeError) instead of waiting in the all-reduce
        from . import png_ring
        with _RING_LOCK:
            loader = _RING_PREFETCH.pop(path, None)
            if loader is None and shard:
                engine = _engine_for(model, dims)
                loader = png_ring.PngRingLoader(shard, batch_size, engine.device, group=device_batch_images(batch_size) // batch_size,
                                                workers=num_workers, start=True)
        if loader is not None:
            loader.device = _engine_for(model, dims).device
        t0 = time.perf_counter()
        from .hostinfo import cfs_throttle
        thr0 = cfs_throttle()
        err, out = None, None
        try:
            out = calculate_activation_statistics(loader if loader is not None else [], model, batch_size, dims, cuda)
        except png_ring.RaggedImages as e:
            if world > 1:
                raise RuntimeError(f"--png-feed ring under torchrun needs images of one size ({e}); use --png-feed dataloader") from e
            err = e
        finally:
            if loader is not None:
                loader.close()                             # stops and joins the feeder thread before the ring is unregistered
        if err is None:
            wall = time.perf_counter() - t0
            if tdist.is_main() and len(shard):
                sec = loader.steady_seconds()
                steady = f"; after the first device batch {(len(shard) - loader.first_item_rows) / sec:.0f} images/s" if sec else ""
                dec = f", all decoded {loader.decode_seconds:.2f} s after the workers started" if loader.decode_seconds else ""
                thr1 = cfs_throttle()
                if os.environ.get("TISE_ALLOC_TRACE") == "1":               # probe: how much of the loop went into hipMalloc (caching allocator misses)
                    st = torch.cuda.memory_stats()
                    dec += (f"; allocator: {st.get('num_device_alloc', 0)} device allocations, {st.get('reserved_bytes.all.peak', 0) / 2**30:.1f} GiB reserved, "
                            f"{st.get('num_alloc_retries', 0)} retries")
                dec += (f"; feeder waited {loader.wait_decode_seconds:.2f} s for decode, {loader.wait_buffer_seconds:.2f} s for a device buffer, "
                        f"{loader.wait_copy_seconds + loader.enqueue_seconds:.2f} s on copies; cgroup CPU throttling during the loop: "
                        f"{thr1[0] - thr0[0]} periods, {(thr1[1] - thr0[1]) / 1e3:.0f} ms")
                print(f"[tise] png feed: {len(shard)} images in {wall:.2f} s ({len(shard) / wall:.0f} images/s on this rank{steady}; "
                      f"{loader.workers} decode processes -> shared pinned ring{dec}; loader batch {batch_size}, device batch "
                      f"{device_batch_images(batch_size)})", file=sys.stderr)
            return out
        print(f"[tise] png feed: images of different sizes ({err}); falling back to the DataLoader path", file=sys.stderr)
    dl_workers = min(num_workers, 32)
    dataset = img_data.Dataset(path, transform=None, file_names=shard)
    dataloader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False, drop_last=True,
                                             num_workers=dl_workers, collate_fn=img_data.collate_u8,
                                             pin_memory=True, worker_init_fn=img_data.worker_init)
    t0 = time.perf_counter()
    out = calculate_activation_statistics(dataloader, model, batch_size, dims, cuda)
    wall = time.perf_counter() - t0
    if tdist.is_main() and len(shard):
        print(f"[tise] png feed: {len(shard)} images in {wall:.2f} s ({len(shard) / wall:.0f} images/s on this rank, {dl_workers} "
              f"DataLoader decode workers, loader batch {batch_size}, device batch {device_batch_images(batch_size)})", file=sys.stderr)
    return out


def _build_model(dims, weights, num_classes, seed):
    block_idx = InceptionV3.BLOCK_INDEX_BY_DIM[dims]
    model = InceptionV3([block_idx], weights=weights, num_classes=num_classes, seed=seed)
    from .inception import to_device_flat
    to_device_flat(model, torch.device("cuda", torch.cuda.current_device()))     # model.cuda() (fid_score.py:232-233) in one copy
    return model


@contextlib.contextmanager
def _own_model(dims, weights, num_classes, seed):
    """The model of ONE call of this module's path-level functions.

    _engine_for hangs the engine on the model and the engine holds the model: a reference cycle, so a finished call's ~1.4 GiB of
    device memory (weights, packed weights, statistics and staging buffers) stays until the garbage collector's next full pass
    (tools/soak_cli_loop.py: 16 calls in one process held 20 GiB; the reference's model dies when calculate_fid_given_paths
    returns, fid_score.py:229-238).  TISE_RELEASE_MODEL=1 cuts the cycle on the way out, and memory then stays flat
    (tests/test_gpu_pipeline.py::test_repeated_fid_calls_in_one_process_release_their_model).  It is NOT the default: with the
    prompt release in place two of thirteen runs of tests/test_gpu_pipeline.py failed in tests that had never failed before
     and the cause was not found in what was left of round 6 -- the collector-driven release is the
    behaviour of rounds 1-5."""
    model = _build_model(dims, weights, num_classes, seed)
    try:
        yield model
    finally:
        if os.environ.get("TISE_RELEASE_MODEL", "0") == "1":
            try:
                model._tise_engine = None
            except Exception:                                              # noqa: BLE001
                pass


def calculate_fid_given_paths(paths, batch_size, cuda, dims, weights=None, num_classes=1000, seed=0,
                              save_stats="", num_workers=8, u8_cache=False):
    """Calculates the FID of two paths (fid_score.py:223-238).  ``weights=None`` = seeded stand-in parameters (the
    CLI only allows that behind --synthetic-weights)."""
    for p in paths:
        if not os.path.exists(p):
            raise RuntimeError("Invalid path: %s" % p)    # :225-227
    _check_cuda(cuda)
    if not u8_cache:
        # the decode workers of the first directory start now: PNG decode overlaps building the model (weights, BatchNorm
        # folding, split packing); the second directory's start when the first one's last chunk is decoded
        first = next((p for p in paths if os.path.isdir(p)), None)
        rest = [p for p in paths if os.path.isdir(p) and p != first]
        ld = prefetch_png_ring(first, batch_size, num_workers) if first else None
        if ld is not None and rest:
            ld.on_all_decoded = lambda: prefetch_png_ring(rest[0], batch_size, num_workers)
    t = _timing("decode workers started, building the model")
    if os.environ.get("TISE_PREALLOC_GB"):                 # probe (tools/cli_child_probe.py): what does the first big hipMalloc of a process cost?
        tp = time.perf_counter()
        _x = torch.empty(int(float(os.environ["TISE_PREALLOC_GB"]) * 2**30), dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        print(f"[tise timing] prealloc {os.environ['TISE_PREALLOC_GB']} GB: {time.perf_counter() - tp:.3f} s", file=sys.stderr, flush=True)
        del _x
    with _own_model(dims, weights, num_classes, seed) as model:
        _engine_for(model, dims)                               # fold BatchNorm, pack the split weights, load the code objects
        t = _timing("model + engine ready", t)
        m1, s1 = _compute_statistics_of_path(paths[0], model, batch_size, dims, cuda, num_workers, u8_cache)
        t = _timing("first side done", t)
        # the first side's covariance is complete: factor it on a side stream while the second side's images are decoded
        # and pushed through the network (Tr sqrtm(S1 S2) is symmetric in its arguments; csrc/frechet.hip)
        dev = torch.device("cuda", torch.cuda.current_device())
        solver = frechet_solver(dims, dev)
        use_pf = tuple(np.shape(s1)) == (dims, dims) and tuple(np.shape(m1)) == (dims,)
        if use_pf:
            solver.prefactor(torch.as_tensor(np.ascontiguousarray(s1, dtype=np.float64), device=dev))
        m2, s2 = _compute_statistics_of_path(paths[1], model, batch_size, dims, cuda, num_workers, u8_cache)
        t = _timing("second side done", t)
        if save_stats and tdist.is_main():
            np.savez(save_stats, mu=np.asarray(m2), sigma=np.asarray(s2))
        if not use_pf or np.shape(m1) != np.shape(m2) or np.shape(s1) != np.shape(s2):
            return calculate_frechet_distance(m1, s1, m2, s2)     # generic path (shape asserts :149-150 included)
        try:
            res = solver.distance_prefactored(np.atleast_1d(m1), np.atleast_1d(m2), np.atleast_2d(s2))
        except _lib.TiseStatusError:
            # the factor is gone: the process-wide solver of this (dims, device) served another distance / factorisation
            # between prefactor() and here (a model's forward hook, a second thread).  The one-call form needs nothing kept.
            return calculate_frechet_distance(m1, s1, m2, s2)
        if res["flags"] & _lib.TISE_FLAG_NONFINITE:          # :156-160 (eps retry) lives in calculate_frechet_distance
            return calculate_frechet_distance(m1, s1, m2, s2)
        calculate_frechet_distance.last_result = res
        return np.float64(res["fid"])


def save_statistics_of_path(path, out_npz, batch_size, cuda, dims, weights=None, num_classes=1000, seed=0,
                            num_workers=8, u8_cache=False):
    """STATS-ONLY mode: the step BEFORE the reference path -- write the ``.npz {mu, sigma}`` that
    ``_compute_statistics_of_path`` (fid_score.py:200-203) reads (the reference ships such files,
    download_evaluation_data.py:11-12, but no script that makes them).  No Frechet distance is solved."""
    if not os.path.exists(path):
        raise RuntimeError("Invalid path: %s" % path)
    _check_cuda(cuda)
    if not u8_cache:
        prefetch_png_ring(path, batch_size, num_workers)   # decode overlaps building the model
    with _own_model(dims, weights, num_classes, seed) as model:
        mu, sigma = _compute_statistics_of_path(path, model, batch_size, dims, cuda, num_workers, u8_cache)
        if tdist.is_main():
            np.savez(out_npz, mu=np.asarray(mu), sigma=np.asarray(sigma))
        return mu, sigma


def class_of_crop(filename):
    """``{stem}_{class}_{count}.png`` (object_fidelity/crop_object.py:45) -> class name (may contain spaces; the
    image stem may contain underscores, so split from the right)."""
    base = os.path.basename(filename)
    base = base[:base.rfind(".")] if "." in base else base
    parts = base.rsplit("_", 2)
    if len(parts) != 3 or not parts[2].isdigit():
        raise ValueError(f"not a crop file name ({{stem}}_{{class}}_{{k}}.png): {filename}")
    return parts[1]


def _class_statistics(path, model, batch_size, dims, num_workers, owner=None):
    """One pass over a crop directory -> {class: StatsAccumulator}.  Every crop is used (no drop-last: a class is
    not a DataLoader); batches are formed over the walk-ordered list and each row is folded into its class."""
    files = img_data.get_filenames(path)
    classes = [class_of_crop(f) for f in files]
    names = sorted(set(classes))
    rank, world, _ = tdist.env_world()
    lo, hi = tdist.shard_range(len(files), rank, world)
    engine = _engine_for(model, dims)
    accs = {c: device.StatsAccumulator(dims, engine.device) for c in names}
    dataset = img_data.Dataset(path, transform=None, file_names=files[lo:hi])
    loader = torch.utils.data.DataLoader(dataset=dataset, batch_size=batch_size, shuffle=False, drop_last=False,
                                         num_workers=min(32, _num_workers(num_workers, world)), collate_fn=img_data.collate_u8,
                                         pin_memory=True, worker_init_fn=img_data.worker_init)
    # The pool3 rows of the directory stay on the device (8 KB per crop); once the walk is complete they are sorted by class
    # and folded into the 80 accumulators by ONE grouped launch (device.stats_update_grouped): every class's S is
    # read-modify-written once per directory.  Round 4 ran one index_select + one 528-workgroup covariance launch per class
    # present in a DEVICE BATCH, each a 2 x 33.5 MB read-modify-write of its S for a handful of rows.  Very large
    # directories are flushed every FLUSH_ROWS crops.
    FLUSH_ROWS = 1 << 18
    index_of = {c: i for i, c in enumerate(names)}
    acc_list = [accs[c] for c in names]
    kept, kept_rows = [], 0
    base = lo

    def flush():
        nonlocal kept, kept_rows
        if not kept_rows:
            return
        feats_all = torch.cat(kept) if len(kept) > 1 else kept[0]
        first = base - kept_rows
        cidx = np.fromiter((index_of[c] for c in classes[first:base]), dtype=np.int64, count=kept_rows)
        order = np.argsort(cidx, kind="stable")                       # walk order kept inside a class
        counts = np.bincount(cidx, minlength=len(names))
        offsets = np.concatenate([[0], np.cumsum(counts)])
        feats_sorted = feats_all.index_select(0, torch.from_numpy(order).to(feats_all.device))
        device.stats_update_grouped(acc_list, feats_sorted, offsets)
        kept, kept_rows = [], 0

    for batch in coalesce_batches(loader, engine.device, device_batch_images(batch_size)):
        feats = _forward_batch(engine, model, batch)
        kept.append(feats)
        kept_rows += feats.shape[0]
        base += feats.shape[0]
        if kept_rows >= FLUSH_ROWS:
            flush()
    flush()
    engine.check_numerics()                                # split-fp16 range guard, agreed on by all ranks before the reductions
    # every class is OWNED by one rank (``owner``: class -> rank, the same map on every rank and for both image sets):
    # its 33.57 MB buffer is reduced to that rank only, which alone finalises and solves it (calculate_per_class_fid) --
    # 80 classes on 8 GPUs: ten reduces and ten Frechet solves per rank instead of 80 all-reduces and 80 redundant
    # solves on every rank
    me = tdist.rank()
    parked = []
    for c in names:                                        # same class list on every rank (same walk)
        dst = owner[c] if owner is not None else 0
        tdist.reduce_sum_(accs[c].buffer(), dst=dst)
        if owner is not None and dst != me:
            # not this rank's class.  The buffer is the library's own hipMalloc memory, not the caching allocator's: it must
            # outlive the asynchronous reduce without relying on hipFree's implicit device-wide synchronisation (which would
            # also drain every reduce before the next is enqueued) -- so the accumulators are parked and dropped together
            parked.append(accs[c])
            accs[c] = None
    if parked:
        torch.cuda.current_stream(engine.device).synchronize()
        parked.clear()
    return accs


def calculate_per_class_fid(paths, batch_size, cuda, dims, weights=None, num_classes=80, seed=0, num_workers=8,
                            min_count=2):
    """EXTENSION (BASELINE configs[4]; the reference computes ONE O-FID over all crops): a Frechet
    distance per object class.  Returns (OrderedDict class -> fid, skipped) where ``skipped`` lists classes with
    fewer than ``min_count`` crops on either side (covariance undefined)."""
    from collections import OrderedDict
    for p in paths:
        if not os.path.isdir(p):
            raise RuntimeError("Invalid path: %s" % p)
    _check_cuda(cuda)
    with _own_model(dims, weights, num_classes, seed) as model:
        # the class list comes from the file names alone: the sorted union over both directories, identical on every rank,
        # dealt round-robin to the ranks
        present = [set(class_of_crop(f) for f in img_data.get_filenames(p)) for p in paths]
        names = sorted(present[0] | present[1])
        world, me = tdist.world_size(), tdist.rank()
        owner = tdist.class_owners(names, world)
        a1 = _class_statistics(paths[0], model, batch_size, dims, num_workers, owner)
        a2 = _class_statistics(paths[1], model, batch_size, dims, num_workers, owner)
        dev = torch.device("cuda", torch.cuda.current_device())
        mine = []
        status = torch.zeros((len(names), 3), dtype=torch.float64, device=dev)       # [fid, solved, skipped] per class
        for i, c in enumerate(names):
            if owner[c] != me:
                continue
            if c not in a1 or c not in a2 or a1[c].count() < min_count or a2[c].count() < min_count:
                status[i, 2] = 1.0
                continue
            mine.append((i, c))
        fids = _solve_classes([(a1[c], a2[c]) for _, c in mine], dims, dev)
        for (i, _), v in zip(mine, fids):
            status[i, 0], status[i, 1] = v, 1.0
        tdist.all_reduce_sum_(status)                                                 # 80 x 3 doubles: every rank gets every class
        st = status.cpu().numpy()
        out, skipped = OrderedDict(), []
        for i, c in enumerate(names):
            if st[i, 1] > 0:
                out[c] = float(st[i, 0])
            else:
                skipped.append(c)
        return out, skipped


_CLASS_SOLVERS = {}


def _solve_classes(pairs, dims, dev, eps=1e-6):
    """Frechet distances of [(StatsAccumulator side 1, side 2), ...] on this rank.  A solve at d = 2048 is ~4 000 short
    dependent launches (csrc/frechet.hip: one per column of the tridiagonalisation), i.e. launch-latency bound, so several
    host threads drive one solver and one stream each and the device interleaves them (TISE_PERCLASS_STREAMS, default 4;
    tools/perclass_probe.py, profiles/r04f_perclass_probe.txt: 80 full-rank solves 2.10 s on one stream, 1.38 / 1.22 / 1.14 s
    on 2 / 3 / 4; 80 rank-300 solves 0.57 -> 0.18 s)."""
    import threading
    n = len(pairs)
    out = [None] * n
    if n == 0:
        return out
    nthr = max(1, min(int(os.environ.get("TISE_PERCLASS_STREAMS", "4")), n))
    main_stream = torch.cuda.current_stream(dev)
    done = torch.cuda.Event()
    done.record(main_stream)
    errs = []

    def work(t):
        try:
            torch.cuda.set_device(dev)
            # one solver + stream per worker slot, kept for the life of the process like engine.frechet_solver: creating them
            # per call cost several d x d hipMallocs each, and closing a solver (hipFree = device-wide synchronisation) while
            # the sibling threads were still enqueueing their ~4000 launches per solve stalled the interleaving
            if nthr > 1:
                key = (int(dims), str(dev), t)
                if key not in _CLASS_SOLVERS:
                    _CLASS_SOLVERS[key] = (device.FrechetSolver(dims, dev), torch.cuda.Stream(device=dev))
                solver, stream = _CLASS_SOLVERS[key]
            else:
                solver, stream = frechet_solver(dims, dev), main_stream
            with torch.cuda.stream(stream):
                stream.wait_event(done)
                for j in range(t, n, nthr):
                    m1, s1 = pairs[j][0].finalize()
                    m2, s2 = pairs[j][1].finalize()
                    res = solver.distance(m1, s1, m2, s2, 0.0)
                    if res["flags"] & _lib.TISE_FLAG_NONFINITE:                    # fid_score.py:156-160
                        print(("fid calculation produces singular product; " "adding %s to diagonal of cov estimates") % eps)
                        res = solver.distance(m1, s1, m2, s2, float(eps))
                    out[j] = float(res["fid"])
                stream.synchronize()
        except BaseException as e:                                                 # noqa: BLE001 -- re-raised below
            errs.append(e)

    if nthr == 1:
        work(0)
    else:
        threads = [threading.Thread(target=work, args=(t,), name=f"tise-frechet-{t}") for t in range(nthr)]
        for th in threads:
            th.start()
        for th in threads:
            th.join()
    if errs:
        raise errs[0]
    return out


def main(argv=None):
    parser = _build_parser()
    args = parser.parse_args(argv)
    _timing(f"imports done (this module's imports {_T_IMPORT1 - _T_IMPORT0:.2f} s)")
    if args.gpu == "":
        _check_cuda(False)
    if args.path1 is None and not args.save_stats:
        parser.error("--path1 is required (omit it only together with --save-stats: statistics-only mode)")
    rank, world, local_rank = tdist.init_from_env()
    if world == 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", args.gpu)        # reference: CUDA_VISIBLE_DEVICES = args.gpu (:243)
    _PNG_FEED["mode"] = args.png_feed
    if args.conv is not None:
        os.environ["TISE_CONV"] = "miopen" if args.conv == "exact" else "split"
    kind = "inception80" if (args.label == "O-FID" and args.num_classes == 80) else "inception"
    wpath, tag = tweights.resolve(args.weights, args.synthetic_weights, kind)
    if args.path1 is None:                                             # statistics-only
        if tdist.is_main():
            print([args.path2])
        mu, sigma = run_with_exact_fallback(lambda: save_statistics_of_path(
            args.path2, args.save_stats, args.batch_size, args.gpu, args.dims, wpath, args.num_classes, args.seed, args.num_workers,
            args.u8_cache), "the statistics pass")
        if tdist.is_main():
            print(f"statistics of {args.path2} -> {args.save_stats}{tag}")
        return None
    paths = [args.path1, args.path2]
    if tdist.is_main():
        print(paths)                                                   # :247
    if args.per_class:
        per, skipped = run_with_exact_fallback(lambda: calculate_per_class_fid(
            paths, args.batch_size, args.gpu, args.dims, wpath, args.num_classes, args.seed, args.num_workers), "the per-class FID")
        mean = float(np.mean(list(per.values()))) if per else float("nan")
        if tdist.is_main():
            lines = [f"{args.label}[{c}]: {v}{tag}" for c, v in per.items()]
            lines.append(f"{args.label} (mean of {len(per)} classes): {mean}{tag}")
            if skipped:
                lines.append("skipped (fewer than 2 crops on a side): " + ", ".join(skipped))
            if args.saved_file:
                with open(args.saved_file, "w") as f:
                    f.write("\n".join(lines))
            print("\n".join(lines))
        return per
    fid_value = run_with_exact_fallback(lambda: calculate_fid_given_paths(
        paths, args.batch_size, args.gpu, args.dims, wpath, args.num_classes, args.seed, args.save_stats, args.num_workers,
        args.u8_cache), "the FID").item()
    if tdist.is_main():
        if args.saved_file:
            with open(args.saved_file, "w") as f:
                f.write(f"{args.label}: {fid_value}{tag}")             # :251-252 (no trailing newline)
        print(f"{args.label}: {fid_value}{tag}")                       # :254
    return fid_value


if __name__ == "__main__":
    tdist.run_cli(main)
