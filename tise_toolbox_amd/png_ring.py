"""PNG feed of the drop-in CLIs: decode workers -> ONE shared, page-locked uint8 ring -> side-stream H2D -> device batches.

Replaces (row a2 of SURVEY section 8) the hand-over of image_realism/FID/fid_score.py:215-217 --
``DataLoader(dataset, batch_size, drop_last=True, num_workers=8)`` whose workers run ``Dataset.__getitem__``
(img_data.py:19-25: ``Image.open(f).convert("RGB")`` + transform) and ship every batch to the main process through a
pickling queue, where it is collated and pinned.  Round 4 measured that hand-over as THE limit of the drop-in: 3 850 images/s
with 8 workers, 4 699 with 32, 3 565 with 64 on a 256-thread host -- falling with more workers, i.e. bound by the parent
(queues + collate + the pin-memory thread), not by zlib (profiles/r04h_cli_host_inclusive.txt).

Here the parent touches no pixel:
  * workers are stand-alone programs started with subprocess -- no torch import, no fork of a process that holds a GPU context,
    no multiprocessing machinery: since round 6 the native ``tise_png_worker`` (csrc/png_worker.c, up in ~2 ms), with
    ``_png_worker.py`` (numpy + Pillow only) as the fallback for chunks that hold a file outside the native decoder's subset
    -- that decode walk-ordered chunks of ``chunk`` files straight into slots of an anonymous shared-memory ring (memfd,
    inherited file descriptors) and set one ``done`` byte per chunk; chunks are claimed under a POSIX record lock, a slot is
    rewritten only after the parent has copied the chunk that was in it (``consumed`` counter);
  * the parent page-locks the ring ONCE (tise_host_register; the ring is parked for the next loader of the process) and a
    feeder thread enqueues ``tise_memcpy_h2d_async`` copies of finished chunks -- one copy per run of chunks in consecutive
    slots --, in order, on THE feed stream of the device (device.feed_stream) into one of three device buffers; a device
    batch is handed to the consumer with an event its stream waits on -- the same contract as img_data.U8CacheLoader
    (``len()`` counts ``batch_size`` batches: the drop-last bookkeeping of fid_score.py:90-96; ``pregrouped``).
Workers can be started BEFORE the model is built (``start()``), so decoding overlaps the process's start-up.
Round 6 (row a2's arithmetic on the GPU): with a device consumer the workers only INFLATE a file -- a ring slot holds the
FILTERED scanlines behind a 64-byte header (csrc/png_decode.c: tise_png_inflate_slot) -- and the five PNG row filters and
the RGBA -> RGB drop run in HBM (csrc/png_unfilter.hip: tise_png_unfilter_rgb8, on the feed's side stream after a device
batch's copies); TISE_PNG_UNFILTER=host keeps the whole decode in the workers, and so does a host consumer (iter_host).
Device batches follow engine.item_schedule (short first and last batches) and are clamped to engine.STAGING_BYTES_CAP of
pixels for large images.
All images must have the size of the first one; a different size raises ``RaggedImages`` and the caller falls back to the
DataLoader path (ragged crop directories are O-FID's, which keeps that path).
"""
import mmap
import os
import queue
import subprocess
import sys
import threading
import time
from collections import deque

import numpy as np
import torch

from . import _lib
from ._png_worker import (DONE_FAILED, DONE_OK, ERRTXT_BYTES, HDR_CHUNK, HDR_CONSUMED, HDR_DONE_OFF, HDR_ERR, HDR_ERRTXT_OFF, HDR_FILES_OFF, HDR_H,
                          HDR_IMG_BYTES, HDR_NCHUNKS, HDR_NEED_PY, HDR_NEXT, HDR_NFILES, HDR_NSLOTS, HDR_RGBONLY, HDR_STARTED, HDR_STOP, HDR_W,
                          HDR_WORDS)

_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_png_worker.py")
_NATIVE_WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tise_png_worker")     # csrc/png_worker.c (build.build_png)


def native_worker():
    """Path of the native decode program, or None (not built, TISE_PNG_WORKER=python, or TISE_PNG_DECODER=pillow -- every
    file through Pillow -- asked for the Python workers)."""
    if os.environ.get("TISE_PNG_WORKER", "native") == "python" or os.environ.get("TISE_PNG_DECODER", "native") == "pillow":
        return None
    return _NATIVE_WORKER if os.access(_NATIVE_WORKER, os.X_OK) else None


class RaggedImages(ValueError):
    """The directory holds images of different sizes (the ring has one slot shape) or, with ``rgb_only``, files that are not
    plain 3-channel RGB: the caller falls back to its DataLoader path."""


from .hostinfo import usable_cpus  # noqa: E402,F401  (affinity and cgroup CPU quota)


def auto_workers(world=1):
    """Decode processes of one rank: the CPUs the process may really use (usable_cpus: affinity and cgroup quota; at most
    128) minus a reserve for the ranks' own threads (launching kernels, the feeder), shared by the ranks of the node, at
    least 2.  Under a CFS quota every runnable thread beyond it gets the WHOLE cgroup throttled in bursts -- the kernel
    launches included; measured on this pool's boxes (16 CPUs of quota; tools/png_feed_probe.py,
    profiles/r06c_png_feed_timeline.txt): 16 / 14 / 12 / 10 / 8 inflate-only workers feed 23.2 / 23.1 / 23.0 / 22.4 / 20.5 k
    images/s of a 25 k job, with two runs at 14-15 workers collapsing to 16 k when the box throttled.
    On these boxes: 14 at world 1, 7 / 3 / 2 at world 2 / 4 / 8."""
    n = min(128, usable_cpus())
    return max(2, (n - max(2, n // 8)) // max(1, world))


def unfilter_on_device(device):
    """Where the PNG row filters are reversed for a consumer on ``device``: on the GPU (default for a HIP device) unless
    TISE_PNG_UNFILTER=host; a host consumer (iter_host, the u8 cache build) always gets pixels."""
    return torch.device(device).type == "cuda" and os.environ.get("TISE_PNG_UNFILTER", "device") != "host"


# One pixel ring is kept alive between loaders of a process (the two directories of an FID run, repeated evaluations): page-locking
# 57 MB costs ~8 ms (hipHostRegister) and the first copy out of freshly registered pages another ~5 ms
# (tools/ring_setup_probe.py, profiles/r06c_png_feed_timeline.txt) -- more than the decode of the first device batch.
_RING_POOL = {"fd": -1, "map": None, "size": 0, "registered_on": None}
_POOL_LOCK = threading.Lock()


def _pool_take(size):
    """(fd, mmap, device it is page-locked for | None) of a parked ring of exactly ``size`` bytes, or a fresh one."""
    with _POOL_LOCK:
        if _RING_POOL["map"] is not None and _RING_POOL["size"] == size:
            out = (_RING_POOL["fd"], _RING_POOL["map"], _RING_POOL["registered_on"])
            _RING_POOL.update(fd=-1, map=None, size=0, registered_on=None)
            return out
    fd = os.memfd_create("tise_png_ring")
    os.ftruncate(fd, size)
    return fd, mmap.mmap(fd, size), None


def _pool_release(fd, m, registered_on):
    """Unregister, unmap and close a ring."""
    if m is None:
        return
    if registered_on is not None:
        try:
            addr = np.frombuffer(m, dtype=np.uint8).ctypes.data
            with torch.cuda.device(registered_on):
                _lib.call("tise_host_unregister", addr)
        except Exception:                                                     # noqa: BLE001
            pass
    try:
        m.close()
    except (BufferError, ValueError):                                         # a numpy view still alive: the fd close below frees it with the process
        pass
    if fd >= 0:
        os.close(fd)


def _pool_park(fd, m, size, registered_on):
    """Keep this ring for the next loader (one ring at most: whatever was parked before is released)."""
    with _POOL_LOCK:
        old = (_RING_POOL["fd"], _RING_POOL["map"], _RING_POOL["registered_on"])
        _RING_POOL.update(fd=fd, map=m, size=size, registered_on=registered_on)
    _pool_release(*old)


_POOL_PID = os.getpid()


def _pool_drop():
    if os.getpid() != _POOL_PID:                                              # a forked child: the parked ring is the parent's
        return
    with _POOL_LOCK:
        old = (_RING_POOL["fd"], _RING_POOL["map"], _RING_POOL["registered_on"])
        _RING_POOL.update(fd=-1, map=None, size=0, registered_on=None)
    _pool_release(*old)


import atexit  # noqa: E402
atexit.register(_pool_drop)
# A process that fork()s while it holds page-locked (hipHostRegister'ed) memory gets children that crash inside the HIP runtime
# (measured: DataLoader workers forked after a ring feed died with SIGSEGV -- tests/test_gpu_clip.py).  The parked ring is
# therefore released before any fork of this interpreter (DataLoader workers, multiprocessing); subprocess launches -- the
# decode workers -- do not run these hooks and do not need them.
if hasattr(os, "register_at_fork"):
    os.register_at_fork(before=_pool_drop)


class PngRingLoader:
    NBUF = 3
    pregrouped = True

    def __init__(self, files, batch_size, device, group=1, workers=None, chunk=8, start=True, rgb_only=False):
        self.files = list(files)
        self.bs = int(batch_size)
        self.group = max(1, int(group))
        self.device = torch.device(device)
        self.n_rows = (len(self.files) // self.bs) * self.bs                 # whole batches only (fid_score.py:90-96)
        self.files = self.files[:self.n_rows]
        self.workers = int(workers) if workers else auto_workers()
        self.chunk = max(1, int(chunk))
        self.procs, self.ring, self.ctl = [], None, None
        self.ring_fd = self.ctl_fd = -1
        self.registered, self.registered_on = False, None
        self.decode_seconds = None
        self.first_item_event = self.last_item_event = None
        self.first_item_rows = 0
        self.t_started = None
        self.on_all_decoded = None                                            # hook: called once when the last chunk is in the ring
        self.rgb_only = bool(rgb_only)                                        # RGBA / palette / gray files are refused (RaggedImages) instead of converted
        self.framed = unfilter_on_device(self.device)                         # slots = [header | filtered rows]: the GPU reverses the filters
        self.feeder = None
        self.wait_decode_seconds = self.wait_buffer_seconds = 0.0             # feeder thread: waiting for a decoded chunk / for a free device buffer
        self.enqueue_seconds = self.wait_copy_seconds = 0.0                   # ... inside hipMemcpyAsync / waiting for copies to land (slot release)
        self.copies_enqueued = 0
        self._pid = os.getpid()                                               # the process that owns the ring, the workers and the HIP registration
        self.py_procs = []                                                    # Python fallback workers (started when a native worker hands a chunk back)
        self.native = None
        if self.n_rows and start:
            self.start()

    def __len__(self):
        return self.n_rows // self.bs

    # ---- shared memory + workers ------------------------------------------------------------------------------------
    def start(self):
        if self.procs or not self.n_rows:
            return self
        from PIL import Image
        with Image.open(self.files[0]) as im:
            w, h = im.size
        self.h, self.w = h, w
        img_bytes = h * w * 3
        if self.framed:
            from . import _png_worker
            lib = _png_worker.load_decoder()
            bpp = 0
            if lib is not None:
                import ctypes
                with open(self.files[0], "rb") as fh_:
                    blob = fh_.read()
                gw, gh, pc = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
                if lib.tise_png_probe(blob, len(blob), ctypes.byref(gw), ctypes.byref(gh), ctypes.byref(pc)) == 0:
                    bpp = pc.value                                             # 3 or 4: the slots are sized for the first file's format
            if lib is None or bpp == 0 or w * bpp + 1 > 8192:
                self.framed = False                                            # no native decoder / a first file outside its subset / rows beyond the kernel's tile
            else:
                img_bytes = int(lib.tise_png_slot_bytes(h, w, bpp))
        n_chunks = -(-self.n_rows // self.chunk)
        self.workers = max(1, min(self.workers, n_chunks))
        # slots: two per worker (one being written, one waiting for its copy), capped at 1 GiB of pinned pixels
        nslots = max(2, min((2 * self.workers + 4) * max(1, int(os.environ.get("TISE_RING_SLOTS_MULT", "1"))), n_chunks,
                            max(2, (1 << 30) // (self.chunk * img_bytes))))
        self.nslots, self.n_chunks, self.img_bytes = nslots, n_chunks, img_bytes
        names = [f.encode("utf-8", "surrogateescape") for f in self.files]
        offs = np.zeros(len(names) + 1, dtype=np.int64)
        np.cumsum([len(b) for b in names], out=offs[1:])
        done_off = 8 * HDR_WORDS
        err_off = done_off + ((n_chunks + 7) & ~7)
        files_off = err_off + ERRTXT_BYTES
        self.ctl_size = files_off + 8 * (len(names) + 1) + int(offs[-1]) + 8
        self.ring_size = nslots * self.chunk * img_bytes
        self.ctl_fd = os.memfd_create("tise_png_ctl")
        os.ftruncate(self.ctl_fd, self.ctl_size)
        self.ctl = mmap.mmap(self.ctl_fd, self.ctl_size)
        self.ring_fd, self.ring, self.registered_on = _pool_take(self.ring_size)
        self.registered = self.registered_on is not None
        self.hdr = np.frombuffer(self.ctl, dtype=np.int64, count=HDR_WORDS)
        self.hdr[:] = 0
        self.hdr[HDR_NCHUNKS], self.hdr[HDR_CHUNK], self.hdr[HDR_NSLOTS] = n_chunks, self.chunk, nslots
        self.hdr[HDR_H], self.hdr[HDR_W], self.hdr[HDR_NFILES] = h, w, len(names)
        self.hdr[HDR_FILES_OFF], self.hdr[HDR_DONE_OFF], self.hdr[HDR_ERRTXT_OFF] = files_off, done_off, err_off
        self.hdr[HDR_RGBONLY] = 1 if self.rgb_only else 0
        self.hdr[HDR_IMG_BYTES] = img_bytes
        self.done = np.frombuffer(self.ctl, dtype=np.uint8, count=n_chunks, offset=done_off)
        np.frombuffer(self.ctl, dtype=np.int64, count=len(names) + 1, offset=files_off)[:] = offs
        blob_off = files_off + 8 * (len(names) + 1)
        self.ctl[blob_off:blob_off + int(offs[-1])] = b"".join(names)
        self.native = native_worker()
        self.t_started = time.perf_counter()
        if self.native is not None:
            # first-line workers: the native program (up in ~2 ms; csrc/png_worker.c).  Chunks it hands back (a file outside
            # its subset) are redone by Python workers started on demand (_start_fallback)
            cmd = [self.native, str(self.ring_fd), str(self.ctl_fd), str(self.ring_size), str(self.ctl_size)]
            for _ in range(self.workers):
                self.procs.append(subprocess.Popen(cmd, pass_fds=(self.ring_fd, self.ctl_fd), stdin=subprocess.DEVNULL))
        else:
            for _ in range(self.workers):
                self.procs.append(self._python_worker(fallback=False))
        return self

    def _python_worker(self, fallback):
        cmd = [sys.executable, "-S", _WORKER, str(self.ring_fd), str(self.ctl_fd), str(self.ring_size), str(self.ctl_size)]
        if fallback:
            cmd.append("--fallback")
        env = dict(os.environ)
        # site-packages must stay importable under -S (numpy, Pillow): hand the parent's path over
        env["PYTHONPATH"] = os.pathsep.join(p for p in sys.path if p)
        for k in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
            env[k] = "1"
        return subprocess.Popen(cmd, env=env, pass_fds=(self.ring_fd, self.ctl_fd), stdin=subprocess.DEVNULL)

    def _start_fallback(self):
        """A native worker met a file it does not decode and handed its chunk back (done byte 3, header NEED_PY): start the
        Python workers that redo such chunks with Pillow -- as many as there are native ones (a directory of JPEGs hands
        EVERY chunk back, and the native workers then cost nothing)."""
        if not self.py_procs and self.ctl is not None:
            self.py_procs = [self._python_worker(fallback=True) for _ in range(self.workers)]

    def _error(self):
        o = int(self.hdr[HDR_ERRTXT_OFF])
        raw = bytes(self.ctl[o:o + ERRTXT_BYTES]).split(b"\0", 1)[0].decode("utf-8", "replace")
        if raw.startswith("ValueError: RAGGED "):
            return RaggedImages(raw[len("ValueError: RAGGED "):])
        return RuntimeError(f"png decode worker failed: {raw}")

    def _wait_chunk(self, c, stop):
        """Block until chunk c is decoded (done byte set); raises on a worker error or a dead worker."""
        spins = 0
        while self.done[c] not in (DONE_OK, DONE_FAILED):                     # 0 not decoded yet, 3 handed back, 4 being redone
            if stop.is_set():
                return False
            if self.hdr[HDR_NEED_PY] and not self.py_procs:
                self._start_fallback()
            spins += 1
            if spins % 2000 == 0:                                             # ~ every 0.4 s: is anybody still alive?
                if self.hdr[HDR_ERR]:
                    raise self._error()
                codes = [p.poll() for p in self.procs + self.py_procs]
                pending = self.done[c] not in (DONE_OK, DONE_FAILED)
                # a worker that was killed (out of memory, a signal) may hold a claimed chunk that nobody will ever finish: the others
                # keep running until the ring is full and then wait for this consumer, which waits for that chunk -- so ANY abnormal
                # exit is an error here, not only the death of all of them
                if any(rc not in (None, 0) for rc in codes) and pending:
                    raise RuntimeError("a png decode worker died before the ring was complete "
                                       f"(exit codes {sorted(set(rc for rc in codes if rc is not None))})")
                if all(rc is not None for rc in codes) and pending and not (self.hdr[HDR_NEED_PY] and not self.py_procs):
                    raise RuntimeError("png decode workers exited before the ring was complete "
                                       f"(exit codes {sorted(set(codes))})")
            time.sleep(0.0002)
        if self.done[c] == DONE_FAILED:
            raise self._error()
        return True

    def iter_host(self):
        """Host-side iteration (no GPU): yields (first row, uint8 view (rows, H, W, 3) of the ring slot) chunk by chunk in walk
        order; the view is valid until the next item is requested (its slot is then released to the workers).  Used by
        img_data.build_u8_cache and by the CPU tests of the worker protocol."""
        if not self.n_rows:
            return
        if self.procs and self.framed:
            raise RuntimeError("iter_host needs pixel slots: construct the loader with a host device (or TISE_PNG_UNFILTER=host)")
        self.framed = False
        self.start()
        stop = threading.Event()
        slots = np.frombuffer(self.ring, dtype=np.uint8).reshape(self.nslots, self.chunk, self.h, self.w, 3)
        try:
            for c in range(self.n_chunks):
                self._wait_chunk(c, stop)
                lo, hi = c * self.chunk, min((c + 1) * self.chunk, self.n_rows)
                yield lo, slots[c % self.nslots][:hi - lo]
                self.hdr[HDR_CONSUMED] = c + 1
            self.decode_seconds = time.perf_counter() - self.t_started
        finally:
            del slots
            self.close()

    # ---- iteration: device batches --------------------------------------------------------------------------------------
    def item_sizes(self):
        """Rows of the consecutive device batches: engine.item_schedule over ``group`` loader batches at most, clamped so that
        one staging buffer holds at most STAGING_BYTES_CAP of pixels (the callers size ``group`` for 256 x 256 images; 1024 x
        1024 files get 341-image device batches, not 3 x 9.4 GB of buffers -- ADVICE r5)."""
        from .engine import STAGING_BYTES_CAP, item_schedule
        cap_rows = max(self.bs, STAGING_BYTES_CAP // max(1, self.h * self.w * 3) // self.bs * self.bs)
        return item_schedule(self.n_rows, self.bs, min(self.bs * self.group, cap_rows))

    def __iter__(self):
        if not self.n_rows:
            return
        self.start()
        dev = self.device
        sizes = self.item_sizes()
        starts = [0]
        for r in sizes:
            starts.append(starts[-1] + r)
        nb = len(sizes)
        nbuf = min(self.NBUF, nb)
        max_rows = max(sizes)
        from .device import feed_stream
        side = feed_stream(dev)                                               # one high-priority stream per device (device.feed_stream)
        side.wait_stream(torch.cuda.current_stream(dev))
        bufs = [torch.empty((max_rows, self.h, self.w, 3), dtype=torch.uint8, device=dev) for _ in range(nbuf)]
        # device-unfilter feed: the slots of a device batch land in ONE staging buffer (copies and the unfilter kernel are
        # stream-ordered on the side stream, so it is free again when the next batch's first copy starts)
        raw = torch.empty((max_rows, self.img_bytes), dtype=torch.uint8, device=dev) if self.framed else None
        for t in bufs + ([raw] if raw is not None else []):
            t.record_stream(side)
        ring_np = np.frombuffer(self.ring, dtype=np.uint8)
        ring_addr = ring_np.ctypes.data
        if self.registered and self.registered_on != dev:                     # a parked ring page-locked for another GPU of this process
            with torch.cuda.device(self.registered_on):
                _lib.call("tise_host_unregister", ring_addr)
            self.registered, self.registered_on = False, None
        if not self.registered:
            with torch.cuda.device(dev):
                try:
                    _lib.call("tise_host_register", ring_addr, self.ring_size)
                    self.registered, self.registered_on = True, dev
                except _lib.TiseStatusError as e:                             # not fatal: the copies become synchronous
                    print(f"[tise] png ring: hipHostRegister failed ({e}); host->device copies will be staged", file=sys.stderr)
        ready = [torch.cuda.Event() for _ in range(nbuf)]
        consumed = [torch.cuda.Event() for _ in range(nbuf)]
        handed = [threading.Semaphore(1) for _ in range(nbuf)]
        out = queue.Queue()
        stop = threading.Event()
        img_bytes, chunk = self.img_bytes, self.chunk
        side_h = side.cuda_stream
        run_max = max(1, min(int(os.environ.get("TISE_RING_RUN", "8")), self.nslots // 4))   # chunks per copy at most (A/B: 1 = one copy per chunk)
        ev_min = max(1, int(os.environ.get("TISE_RING_EVENT_CHUNKS", "1")))                  # chunks between two slot-release events at least (A/B)

        def feeder():
            try:
                torch.cuda.set_device(dev)
                inflight = deque()                                            # (event after the chunk's copies, chunk)
                c = 0
                last_copied = last_evented = -1                               # last chunk whose copy is enqueued / covered by an event
                for b in range(nb):
                    k = b % nbuf
                    tw = time.perf_counter()
                    handed[k].acquire()
                    if stop.is_set():
                        return
                    consumed[k].synchronize()                                 # the consumer's stream is done with buffer k
                    self.wait_buffer_seconds += time.perf_counter() - tw
                    r0, r1 = starts[b], starts[b + 1]
                    base = raw.data_ptr() if self.framed else bufs[k].data_ptr()
                    r = r0
                    while r < r1:
                        c = r // chunk
                        tw = time.perf_counter()
                        if not self._wait_chunk(c, stop):
                            return
                        self.wait_decode_seconds += time.perf_counter() - tw
                        # ONE copy for the run of chunks that are decoded already and sit in consecutive ring slots (at most
                        # run_max of them: their slots are released together): the decoders run ahead of this loop most of the
                        # time, and every operation on the feed stream -- copy, event -- has a fixed cost that grows when the host
                        # is busy (round 6: tools/cli_copy_trace.sh -- slow runs had FAST copies that started late)
                        c_last = c
                        while (c_last + 1 - c < run_max and c_last + 1 < self.n_chunks and (c_last + 1) * chunk < r1
                               and (c_last + 1) % self.nslots != 0 and self.done[c_last + 1] == DONE_OK):
                            c_last += 1
                        hi = min((c_last + 1) * chunk, r1, self.n_rows)
                        src = ring_addr + ((c % self.nslots) * chunk + (r - c * chunk)) * img_bytes
                        tw = time.perf_counter()
                        _lib.call("tise_memcpy_h2d_async", base + (r - r0) * img_bytes, src, (hi - r) * img_bytes, side_h)
                        self.enqueue_seconds += time.perf_counter() - tw
                        self.copies_enqueued += 1
                        # chunks whose last row is now on its way: their slots free when this copy lands
                        c_done = c_last if hi == min((c_last + 1) * chunk, self.n_rows) else c_last - 1
                        if c_done >= c:
                            last_copied = c_done
                        # an event (= a barrier packet in the stream's compute queue) only every ev_min chunks, at the end of a device
                        # batch, at the last chunk, or when a quarter of the ring waits for its release
                        if last_copied > last_evented and (last_copied - last_evented >= ev_min or hi == r1 or last_copied == self.n_chunks - 1
                                                            or last_copied + 1 - int(self.hdr[HDR_CONSUMED]) >= max(1, self.nslots // 4)):
                            c_done = last_evented = last_copied
                            ev = torch.cuda.Event()
                            ev.record(side)
                            inflight.append((ev, c_done))
                            if c_done == self.n_chunks - 1:
                                self.decode_seconds = time.perf_counter() - self.t_started
                                if self.on_all_decoded is not None:
                                    self.on_all_decoded()
                        tw = time.perf_counter()
                        # release slots whose copies have landed; never let more than half the ring wait for its release
                        while inflight and (inflight[-1][1] + 1 - int(self.hdr[HDR_CONSUMED]) > self.nslots // 2 or inflight[0][0].query()):
                            ev, cc = inflight.popleft()
                            ev.synchronize()
                            self.hdr[HDR_CONSUMED] = cc + 1
                        self.wait_copy_seconds += time.perf_counter() - tw
                        r = hi
                    if self.framed:                                           # row filters + RGBA -> RGB on the GPU, behind the copies
                        _lib.call("tise_png_unfilter_rgb8", raw.data_ptr(), r1 - r0, img_bytes, self.h, self.w, bufs[k].data_ptr(), side_h)
                    ready[k].record(side)
                    out.put((k, r1 - r0))
                while inflight:
                    ev, cc = inflight.popleft()
                    ev.synchronize()
                    self.hdr[HDR_CONSUMED] = cc + 1
            except BaseException as e:                                        # noqa: BLE001 -- re-raised in the consumer
                out.put(e)

        th = threading.Thread(target=feeder, name="tise-png-feeder", daemon=True)
        self.feeder = (th, stop, handed, side)
        th.start()
        try:
            for b in range(nb):
                item = out.get()
                if isinstance(item, BaseException):
                    raise item
                k, rows = item
                cur = torch.cuda.current_stream(dev)
                if b == 1:
                    self.first_item_rows = sizes[0]
                    self.first_item_event = torch.cuda.Event(enable_timing=True)
                    self.first_item_event.record(cur)
                cur.wait_event(ready[k])
                yield bufs[k][:rows]
                consumed[k].record(torch.cuda.current_stream(dev))
                handed[k].release()
            if self.first_item_event is not None:
                self.last_item_event = torch.cuda.Event(enable_timing=True)
                self.last_item_event.record(torch.cuda.current_stream(dev))
        finally:
            self._stop_feeder()
            self.close()

    def _stop_feeder(self):
        """Stop and join the feeder thread and drain the side stream: nothing may still enqueue copies from the ring, or have
        one in flight, when close() unregisters and unmaps it (a consumer-side exception used to reach close()
        with the generator suspended and the feeder alive)."""
        if self.feeder is None:
            return
        th, stop, handed, side = self.feeder
        self.feeder = None
        stop.set()
        for h in handed:
            h.release()
        th.join()
        side.synchronize()

    def steady_seconds(self):
        """Device time between the end of the first and of the last device batch's work (None with fewer than two)."""
        if self.first_item_event is None or self.last_item_event is None:
            return None
        self.last_item_event.synchronize()
        return self.first_item_event.elapsed_time(self.last_item_event) * 1e-3

    def close(self):
        if os.getpid() != getattr(self, "_pid", os.getpid()):
            # a fork()ed child (a DataLoader worker) that inherited this object: its garbage collector may finalise it -- the ring,
            # the decode processes and the HIP registration belong to the PARENT; touching them here stopped the parent's workers
            # and crashed inside the HIP runtime (SIGSEGV in the worker, tests/test_gpu_clip.py under the full suite)
            return
        self._stop_feeder()
        if self.ctl is not None:
            try:
                self.hdr[HDR_STOP] = 1
            except (ValueError, TypeError):
                pass
        for p in self.procs + self.py_procs:
            try:
                p.wait(timeout=5)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        self.procs, self.py_procs = [], []
        self.hdr = self.done = None
        if self.ctl is not None:
            try:
                self.ctl.close()
            except (BufferError, ValueError):                                 # a numpy view still alive: the fd close below frees it with the process
                pass
        self.ctl = None
        if self.ctl_fd >= 0:
            os.close(self.ctl_fd)
        self.ctl_fd = -1
        if self.ring is not None:
            # every worker has exited and the feeder is joined: nobody writes the ring any more.  A page-locked ring is parked for
            # the next loader of this process (TISE_PNG_RING_POOL=0: released), anything else is released now
            if self.registered and os.environ.get("TISE_PNG_RING_POOL", "1") != "0":
                _pool_park(self.ring_fd, self.ring, self.ring_size, self.registered_on)
            else:
                _pool_release(self.ring_fd, self.ring, self.registered_on if self.registered else None)
        self.ring, self.ring_fd, self.registered, self.registered_on = None, -1, False, None

    def __del__(self):
        try:
            self.close()
        except Exception:                                                     # noqa: BLE001
            pass
