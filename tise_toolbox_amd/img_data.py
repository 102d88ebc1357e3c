"""Image directory walking and decoding (mirror of image_realism/FID/img_data.py:7-35).

Same file-selection rule and order as the reference (``os.walk`` order, unsorted; a file is
kept when its NAME contains "jpg" or "png" anywhere, img_data.py:27-35 -- so ``.jpeg`` is
skipped and ``x.png.txt`` is taken, exactly as upstream).  What differs is what a sample is:
the reference resizes on the CPU worker (PIL) and ships fp32 CHW; here a sample is the
decoded uint8 HWC image and the resize runs on the GPU (``csrc/resize.hip``, bit-exact to
PIL), which cuts the worker->trainer traffic 5.5x (196 608 B vs 1 072 812 B per image).
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils import data


def get_filenames(data_path):
    """img_data.py:27-35 / inception_score_star_coco.py:124-135, verbatim semantics."""
    images = []
    for path, subdirs, files in os.walk(data_path):
        for name in files:
            if name.rfind("jpg") != -1 or name.rfind("png") != -1:
                filename = os.path.join(path, name)
                if os.path.isfile(filename):
                    images.append(filename)
    return images


class Dataset(data.Dataset):
    """Characterizes a dataset for PyTorch (img_data.py:7).  ``transform`` is applied to the
    PIL image when given (reference behaviour); with ``transform=None`` the sample is the
    uint8 HWC tensor the device pipeline expects."""

    def __init__(self, path, transform=None, file_names=None):
        self.file_names = self.get_filenames(path) if file_names is None else list(file_names)
        self.transform = transform

    def __len__(self):
        return len(self.file_names)

    def __getitem__(self, index):
        img = Image.open(self.file_names[index]).convert("RGB")      # img_data.py:21
        if self.transform is not None:
            return self.transform(img)
        return torch.from_numpy(np.asarray(img).copy())              # (H, W, 3) uint8

    def get_filenames(self, data_path):
        return get_filenames(data_path)


def files_fingerprint(files, root=None):
    """sha256 over the walk-ordered list of (path relative to `root`, size, mtime_ns): what a decoded-pixel cache of
    `files` is valid for.  Regenerating images into the same directory -- same names, same count, the
    evaluate-every-checkpoint workflow -- changes mtime (and usually size), so the fingerprint changes."""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        st = os.stat(f)
        rel = os.path.relpath(f, root) if root else f
        h.update(f"{rel}\0{st.st_size}\0{st.st_mtime_ns}\n".encode())
    return h.hexdigest()


def u8_cache_is_current(cache_path, files, root=None):
    """True when `cache_path` and its fingerprint side file exist, the fingerprint matches `files` as they are on disk
    now and the array has one row per file."""
    side = cache_path + ".sha256"
    if not (os.path.exists(cache_path) and os.path.exists(side)):
        return False
    try:
        with open(side) as f:
            want = f.read().strip()
        return want == files_fingerprint(files, root) and np.load(cache_path, mmap_mode="r").shape[0] == len(files)
    except (OSError, ValueError):
        return False


def build_u8_cache(files, out_path, num_workers=8, batch_size=256, root=None):
    """Decode `files` ONCE (walk order kept) into a memory-mappable ``.npy`` of shape (N, H, W, 3) uint8 -- SURVEY H2:
    50 k PNG/s of decode is 50-100 CPU cores, so repeated evaluations of the same image set (and multi-GPU runs) read
    this cache instead: 5.9 GB for 30 k x 256 x 256.  All images must have the same size.
    The array is built in ``<out_path>.tmp.<pid>`` and renamed over `out_path` only after it is complete and flushed
    (an interrupted build never leaves a cache that looks valid); ``<out_path>.sha256`` holds files_fingerprint() taken
    BEFORE decoding, so files that change during the build invalidate the cache on the next run."""
    if not files:
        raise ValueError("no image files")
    fingerprint = files_fingerprint(files, root)
    first = np.asarray(Image.open(files[0]).convert("RGB"))
    h, w, _ = first.shape
    final_path, out_path = out_path, f"{out_path}.tmp.{os.getpid()}"
    side = final_path + ".sha256"
    if os.path.exists(side):
        os.remove(side)                                   # the old fingerprint must not outlive the old array
    arr = np.lib.format.open_memmap(out_path, mode="w+", dtype=np.uint8, shape=(len(files), h, w, 3))
    # decode processes write into a shared ring, this process only copies finished chunks into the file (png_ring.py:
    # no pickling queues; round 4 built the cache through a DataLoader at ~4 k images/s)
    from . import png_ring
    ring = png_ring.PngRingLoader(files, 1, "cpu", workers=max(1, num_workers), start=True)
    i = 0
    try:
        try:
            for lo, view in ring.iter_host():
                arr[lo:lo + view.shape[0]] = view
                i = lo + view.shape[0]
        except png_ring.RaggedImages as e:
            raise ValueError(f"--u8-cache needs images of one size; {e}") from e
        if i != len(files):
            raise RuntimeError(f"--u8-cache: decoded {i} of {len(files)} images")
        arr.flush()
        del arr
        os.replace(out_path, final_path)
        with open(side + ".tmp", "w") as f:
            f.write(fingerprint + "\n")
        os.replace(side + ".tmp", side)
    finally:
        if os.path.exists(out_path):
            os.remove(out_path)
    return final_path


class U8CacheLoader:
    """Batches of a (N, H, W, 3) uint8 ``.npy`` cache, delivered ALREADY ON THE DEVICE.  A feeder thread runs up to two
    batches ahead of the consumer: it reads a batch from the file straight into one of three page-locked buffers
    (``os.preadv`` from four threads: the kernel copies page cache -> pinned memory, no mmap page faults), enqueues the
    host->device copy on a side stream and hands the device buffer over with an event the consumer's stream waits on;
    a buffer is reused once the consumer's stream has passed the point where it was handed back.  Same contract as the
    DataLoader it replaces: ``len()`` = number of batches, drop_last=True semantics (fid_score.py:215-217).  ``rows`` =
    (lo, hi) restricts it to a shard of the cache (data-parallel runs).  ``h2d_seconds``: time the feeder spent reading
    and enqueueing (not waiting for buffers).
    ``group`` = K > 1: the DEVICE batch is decoupled from ``batch_size`` -- an item is up to K consecutive batches
    (engine.item_schedule: short first and last items, equal ones between), read, copied and handed over as one tensor;
    ``len()`` still counts batches of ``batch_size`` (the drop-last bookkeeping of fid_score.py:90-96), ``pregrouped``
    tells the consumer."""

    NBUF = 3
    READERS = 4

    def __init__(self, cache_path, batch_size, device, rows=None, group=1):
        self.path = cache_path
        self.arr = np.load(cache_path, mmap_mode="r")
        if self.arr.ndim != 4 or self.arr.shape[3] != 3 or self.arr.dtype != np.uint8:
            raise ValueError(f"{cache_path}: expected a (N, H, W, 3) uint8 array")
        self.lo, self.hi = rows if rows is not None else (0, self.arr.shape[0])
        self.bs = int(batch_size)
        self.group = max(1, int(group))
        self.pregrouped = self.group > 1
        self.device = torch.device(device)
        self.h2d_seconds = 0.0
        self.first_item_done_at, self.first_item_rows = None, 0
        self.first_item_event, self.last_item_event = None, None

    def __len__(self):
        return (self.hi - self.lo) // self.bs

    def __iter__(self):
        import queue
        import threading
        import time
        from concurrent.futures import ThreadPoolExecutor
        n_rows = len(self) * self.bs                                    # whole batches only
        if n_rows == 0:
            return
        # device batches: the schedule every feed of the CLIs uses (engine.item_schedule: short first and last batches), so
        # that the same files give the same fp64 sums -- the same FID to the last bit -- whichever feed delivered them
        from .engine import item_schedule
        sizes = item_schedule(n_rows, self.bs, self.bs * self.group)
        starts = [0]
        for r_ in sizes:
            starts.append(starts[-1] + r_)
        nb = len(sizes)
        shape = (max(sizes),) + tuple(self.arr.shape[1:])
        row_bytes = int(np.prod(shape[1:]))
        data_offset = int(self.arr.offset)
        nbuf = min(self.NBUF, nb)
        pinned = [torch.empty(shape, dtype=torch.uint8).pin_memory() for _ in range(nbuf)]
        dev = [torch.empty(shape, dtype=torch.uint8, device=self.device) for _ in range(nbuf)]
        views = [memoryview(p.numpy()).cast("B") for p in pinned]
        from .device import feed_stream
        side = feed_stream(self.device)                                 # one high-priority stream per device
        side.wait_stream(torch.cuda.current_stream(self.device))       # the device buffers may be memory that queued kernels still use
        for t in dev:
            t.record_stream(side)                                       # ... and the allocator must not recycle them under a copy still in flight
        ready = [torch.cuda.Event() for _ in range(nbuf)]
        consumed = [torch.cuda.Event() for _ in range(nbuf)]
        handed = [threading.Semaphore(1) for _ in range(nbuf)]        # released when the consumer gave slot k back
        out = queue.Queue()
        stop = threading.Event()
        fd = os.open(self.path, os.O_RDONLY)

        def read_chunk(k, off, lo, hi):
            pos = lo
            while pos < hi:                                              # preadv may return short counts
                got = os.preadv(fd, [views[k][pos:hi]], off + pos)
                if got <= 0:
                    raise IOError(f"{self.path}: short read")
                pos += got

        def feeder():
            try:
                torch.cuda.set_device(self.device)
                with ThreadPoolExecutor(self.READERS) as pool:
                    for b in range(nb):
                        k = b % nbuf
                        handed[k].acquire()                              # the consumer returned slot k ...
                        if stop.is_set():
                            return
                        consumed[k].synchronize()                        # ... and its stream is past that point
                        t0 = time.perf_counter()
                        rows = sizes[b]
                        batch_bytes = rows * row_bytes
                        off = data_offset + (self.lo + starts[b]) * row_bytes
                        step = -(-batch_bytes // self.READERS)
                        list(pool.map(lambda i: read_chunk(k, off, i * step, min(batch_bytes, (i + 1) * step)), range(self.READERS)))
                        with torch.cuda.stream(side):
                            dev[k][:rows].copy_(pinned[k][:rows], non_blocking=True)
                            ready[k].record(side)
                        self.h2d_seconds += time.perf_counter() - t0
                        out.put((k, rows))
            except BaseException as e:                                   # noqa: BLE001 -- re-raised in the consumer
                out.put(e)

        th = threading.Thread(target=feeder, name="tise-u8-feeder", daemon=True)
        th.start()
        try:
            for b in range(nb):
                k = out.get()
                if isinstance(k, BaseException):
                    raise k
                k, rows = k
                cur = torch.cuda.current_stream(self.device)
                if b == 1:
                    # the consumer came back for the second item: the first device batch is ENQUEUED, not finished (the trunk is
                    # asynchronous) -- so the steady-state rate is taken between two device events, not from this host time
                    self.first_item_done_at = time.perf_counter()
                    self.first_item_rows = sizes[0]
                    self.first_item_event = torch.cuda.Event(enable_timing=True)
                    self.first_item_event.record(cur)
                cur.wait_event(ready[k])
                yield dev[k][:rows]
                consumed[k].record(torch.cuda.current_stream(self.device))
                handed[k].release()
            if self.first_item_event is not None:
                self.last_item_event = torch.cuda.Event(enable_timing=True)
                self.last_item_event.record(torch.cuda.current_stream(self.device))
        finally:
            stop.set()
            for h in handed:
                h.release()
            th.join()
            side.synchronize()                                          # an early exit: no copy may outlive the buffers
            os.close(fd)

    def steady_seconds(self):
        """Device time from the end of the first device batch's work to the end of the last one's (None with fewer than two
        items): what the feed sustains without the process's first-use costs.  Synchronises on the last event."""
        if self.first_item_event is None or self.last_item_event is None:
            return None
        self.last_item_event.synchronize()
        return self.first_item_event.elapsed_time(self.last_item_event) * 1e-3


def worker_init(_worker_id):
    """``worker_init_fn`` of every DataLoader of this package.  The workers are FORKED from a process that holds a HIP context:
    objects the parent had not yet collected (an engine and its device handles in a reference cycle, say) are inherited as
    garbage, and the first full collection in the child would run their finalisers there -- hipFree and friends in a process that
    must not touch the parent's context ("DataLoader worker exited unexpectedly", seen once in ~10 runs of the 80-class test).
    ``gc.freeze()`` moves everything allocated before the fork out of the collector's reach for the life of the worker; what the
    worker allocates itself is collected as usual.  (The handle classes of device.py also refuse to destroy from another pid.)"""
    import gc
    gc.freeze()


def collate_u8(samples):
    """Stack equal-sized uint8 images to (B,H,W,3); otherwise keep a list (ragged crops, O-FID)."""
    if all(s.shape == samples[0].shape for s in samples):
        return torch.stack(samples, 0)
    return list(samples)
