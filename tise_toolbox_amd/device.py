"""Torch-tensor front ends of the libtise_hip.so entry points.

PyTorch is only plumbing here (device memory, streams): every function passes raw
``data_ptr()`` values and the current HIP stream through the C ABI; the arithmetic is in
``csrc/*.hip``.  All functions raise if the tensors are not on a GPU -- there is no CPU
path in the product (see ``_lib.TiseLibraryError``).
"""
import ctypes
import os

import numpy as np
import torch

from . import _lib

# image_realism/FID/inception.py:120-124 (applied to [0,1] pixels)
NORM_SCALE = (0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5)
NORM_BIAS = ((0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


_FEED_STREAMS = {}


def _pick_independent_stream(dev):
    """A normal-priority stream whose hardware queue is NOT the one the current stream's kernels sit in, found by trying:
    a ~1.5 ms spin kernel goes to the current stream, one tiny host->device copy to each of eight pool streams, and the
    first stream whose copy completes while the spin is still running is taken (None when none overtakes)."""
    import time
    cur = torch.cuda.current_stream(dev)
    cands = [torch.cuda.Stream(device=dev) for _ in range(8)]
    src = torch.zeros(64, dtype=torch.uint8).pin_memory()
    dst = torch.empty(64, dtype=torch.uint8, device=dev)
    for st in cands:                                           # first use creates a stream's queue (milliseconds): not inside the measurement
        with torch.cuda.stream(st):
            dst.copy_(src, non_blocking=True)
        st.synchronize()
    cur.synchronize()
    torch.cuda._sleep(8_000_000)                               # ~4 ms of cycles on the current stream
    end = torch.cuda.Event()
    end.record(cur)
    evs = []
    for st in cands:
        with torch.cuda.stream(st):
            dst.copy_(src, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(st)
        evs.append(ev)
    chosen, t0 = None, time.perf_counter()
    while chosen is None and not end.query() and time.perf_counter() - t0 < 0.1:
        for i, ev in enumerate(evs):
            if ev.query():
                chosen = i
                break
    end.synchronize()
    for st in cands:
        st.synchronize()
    if os.environ.get("TISE_FEED_DEBUG") == "1":
        import sys
        print(f"[tise] feed stream probe: stream {chosen} of {len(cands)} overtook the compute stream", file=sys.stderr)
    return cands[chosen] if chosen is not None else None


def feed_stream(dev):
    """THE side stream of the image feeds on ``dev`` (host->device copies of decoded chunks, the PNG unfilter kernel, the
    gather copies of coalesce_u8): one per device and process.

    Round 6 measurement (tools/png_feed_probe.py, profiles/r06c_png_feed_timeline.txt): a fresh ``torch.cuda.Stream()`` per
    loader comes from torch's round-robin pool and HIP maps streams onto four hardware queues; every FOURTH stream shares
    its queue with the stream the trunk runs on, and its 1 500 chunk copies (1.6 MB each) then queued BEHIND the convolution
    launches -- 451 ms waiting for copies in a 470 ms job, 16 k images/s instead of 23 k, for every second loader of the
    probe.  A HIGH-priority stream always has a queue of its own, but its operations hold the compute queue up while they
    run: bench.py's host_feed leg (600 gather copies of 9.8 MB) fell from 0.98 to 0.89 of the resident rate
    (tools/host_feed_ab.sh, profiles/r06i_feed_stream_ab.txt).  So: a normal-priority stream that is PROVEN to overtake a
    kernel on the current stream (_pick_independent_stream, ~2 ms once per process); the high-priority stream only when
    none does.  TISE_FEED_PRIORITY=high | normal forces either without the probe."""
    dev = torch.device(dev)
    key = (dev.index if dev.index is not None else torch.cuda.current_device())
    st = _FEED_STREAMS.get(key)
    if st is None:
        lo, hi = torch.cuda.Stream.priority_range()            # (lowest, highest) = (0, -1) on this runtime
        mode = os.environ.get("TISE_FEED_PRIORITY", "auto")
        if mode == "auto":
            st = _pick_independent_stream(dev)
        elif mode == "normal":
            st = torch.cuda.Stream(device=dev)
        if st is None:
            st = torch.cuda.Stream(device=dev, priority=hi)
        _FEED_STREAMS[key] = st
    return st


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr())


def _require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.TiseLibraryError("tise_toolbox_amd runs on MI355X only: tensor is not on a HIP device")


def make_lut(normalize_input=True, scale_pm1=False):
    """3x256 fp32 table: byte -> network input value, with the reference's own op order.

    ToTensor (fid_score.py:211): fp32(v) / 255 (true division, fp32).  Then either the
    inception.py:120-124 affine ``x * (s/0.5) + (m-0.5)/0.5`` (fp32 multiply, fp32 add, Python
    scalars rounded to fp32 first, as torch does for tensor-scalar ops) or, for O-IS,
    Normalize((.5,.5,.5),(.5,.5,.5)) = (x - 0.5) / 0.5 (object_centric_inception_score.py:91).
    """
    v = np.arange(256, dtype=np.float32) / np.float32(255.0)
    lut = np.empty((3, 256), dtype=np.float32)
    for c in range(3):
        if scale_pm1:
            lut[c] = (v - np.float32(0.5)) / np.float32(0.5)
        elif normalize_input:
            lut[c] = v * np.float32(NORM_SCALE[c]) + np.float32(NORM_BIAS[c])
        else:
            lut[c] = v
    return np.ascontiguousarray(lut)


_IDENTITY_LUT = np.ascontiguousarray(np.tile(np.arange(256, dtype=np.float32) / np.float32(255.0), (3, 1)))


def resize_bilinear_u8(src_u8, out_hw=(299, 299), lut=None, channels_last=True, return_u8=False):
    """(N,H,W,3) uint8 CUDA tensor -> (N,3,oh,ow) fp32 network input (PIL-exact bilinear).

    Fuses transforms.Resize + ToTensor (fid_score.py:208-213) and the input affine
    (inception.py:120-124) through ``lut`` (see make_lut).  With ``channels_last`` the
    result is an NCHW tensor in torch.channels_last memory format.
    """
    _require_cuda(src_u8)
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3:
        raise ValueError("src_u8 must be (N,H,W,3) uint8")
    src_u8 = src_u8.contiguous()
    n, h, w, _ = src_u8.shape
    oh, ow = out_hw
    if lut is None:
        lut = make_lut(True)
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    if channels_last:
        store = torch.empty((n, oh, ow, 3), dtype=torch.float32, device=src_u8.device)
        out = store.permute(0, 3, 1, 2)
    else:
        store = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=src_u8.device)
        out = store
    u8 = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=src_u8.device) if return_u8 else None
    _lib.call("tise_resize_bilinear_u8", _ptr(src_u8), n, h, w, _ptr(store), oh, ow, 1 if channels_last else 0,
              lut.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ptr(u8) if u8 is not None else None, _stream())
    return (out, u8) if return_u8 else out


def resize_u8_lut(src_u8, out_hw, lut, filter="bicubic", channels_last=False, return_u8=False):
    """(N,H,W,3) uint8 CUDA tensor -> (N,3,oh,ow) fp32 through a 3x256 byte -> value table, with Pillow's 8-bit two-pass
    resample for ``filter`` "bilinear" or "bicubic" (tise_resize_u8).  The CLIP metrics' preprocess on the device:
    clip._transform = Resize(224, BICUBIC) -> CenterCrop (a no-op for square images) -> ToTensor -> Normalize."""
    _require_cuda(src_u8)
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3:
        raise ValueError("src_u8 must be (N,H,W,3) uint8")
    src_u8 = src_u8.contiguous()
    n, h, w, _ = src_u8.shape
    oh, ow = out_hw
    lut = np.ascontiguousarray(lut, dtype=np.float32)
    if channels_last:
        store = torch.empty((n, oh, ow, 3), dtype=torch.float32, device=src_u8.device)
        out = store.permute(0, 3, 1, 2)
    else:
        store = torch.empty((n, 3, oh, ow), dtype=torch.float32, device=src_u8.device)
        out = store
    u8 = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=src_u8.device) if return_u8 else None
    _lib.call("tise_resize_u8", _ptr(src_u8), n, h, w, _ptr(store), oh, ow, 1 if channels_last else 0,
              lut.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ptr(u8) if u8 is not None else None,
              {"bilinear": 0, "bicubic": 1}[filter], _stream())
    return (out, u8) if return_u8 else out


def resize_u8_only(src_u8, out_hw=(299, 299), out=None):
    """(N,H,W,3) uint8 CUDA tensor -> the Pillow-exact resized uint8 image (N,oh,ow,3); no float output (the stem
    convolution applies the input table itself: SplitTrunk.forward_u8).  ``out``: preallocated contiguous
    (N,oh,ow,3) uint8 destination (a slice of a batch buffer when crops of different sizes are stacked)."""
    _require_cuda(src_u8)
    if src_u8.dtype != torch.uint8 or src_u8.dim() != 4 or src_u8.shape[3] != 3:
        raise ValueError("src_u8 must be (N,H,W,3) uint8")
    src_u8 = src_u8.contiguous()
    n, h, w, _ = src_u8.shape
    oh, ow = out_hw
    if out is None:
        u8 = torch.empty((n, oh, ow, 3), dtype=torch.uint8, device=src_u8.device)
    else:
        u8 = out
        if u8.dtype != torch.uint8 or tuple(u8.shape) != (n, oh, ow, 3) or not u8.is_contiguous() or not u8.is_cuda:
            raise ValueError("out must be a contiguous (N,oh,ow,3) uint8 CUDA tensor")
    lut = _IDENTITY_LUT                                                   # unused by the kernel when dst is NULL
    _lib.call("tise_resize_bilinear_u8", _ptr(src_u8), n, h, w, None, oh, ow, 1,
              lut.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), _ptr(u8), _stream())
    return u8


def read_split_overflow():
    """Read-and-clear the range guard of the split-fp16 activation format (csrc/common.h): True when any kernel
    since the last read converted a value above the fp16 range (65504) or a NaN into a split tensor.  Synchronises
    the current stream."""
    flag = ctypes.c_int(0)
    _lib.call("tise_split_overflow_check", ctypes.byref(flag), _stream())
    return bool(flag.value)


def check_split_overflow(what="InceptionV3 trunk", flag=None):
    """Raise when the range guard fired: the hi half would hold +inf and every later layer would be silently wrong."""
    if flag is None:
        flag = read_split_overflow()
    if flag:
        raise FloatingPointError(
            f"{what}: an activation exceeded the fp16 range of the split-precision format (|v| > 65504) or was NaN; "
            "the CLIs finish such a job on the exact-fp32 path by themselves (engine.run_with_exact_fallback); library callers: "
            "TISE_CONV=miopen / --conv exact")


class StatsAccumulator:
    """fp64 running {n, sum x, sum x x^T} on the device (tise_stats_* in include/tise_hip.h)."""

    def __init__(self, dims, device=None):
        self.dims = int(dims)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.TiseLibraryError("StatsAccumulator needs a HIP device")
        self._h = ctypes.c_void_p()
        self._pid = os.getpid()
        with torch.cuda.device(self.device):
            _lib.call("tise_stats_create", self.dims, ctypes.byref(self._h))
        self._keep = []

    def close(self):
        # Only the process that created the handle may destroy it: a forked child (a DataLoader worker of the ragged-crop
        # path) inherits this object, and when ITS garbage collector finalises an unreachable copy the destroy would call
        # hipFree in a process that must not touch the parent's HIP context (crash: "DataLoader worker exited unexpectedly")
        if self._h and os.getpid() == getattr(self, "_pid", None):
            _lib.load().tise_stats_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def reset(self):
        _lib.call("tise_stats_reset", self._h, _stream())

    def update(self, feats):
        """feats: (rows, dims) fp32 CUDA tensor (row stride may exceed dims)."""
        _require_cuda(feats)
        if feats.dim() != 2 or feats.shape[1] != self.dims or feats.dtype != torch.float32:
            raise ValueError(f"feats must be (rows,{self.dims}) float32")
        if feats.stride(1) != 1:
            feats = feats.contiguous()
        _lib.call("tise_stats_update", self._h, _ptr(feats), feats.shape[0], feats.stride(0), _stream())

    def update_parts(self, feats, cov=True, col_sum=True):
        """The two kernels of update() separately (bench.py brackets the MFMA one with HIP events)."""
        if cov:
            _lib.call("tise_stats_update_cov", self._h, _ptr(feats), feats.shape[0], feats.stride(0), _stream())
        if col_sum:
            _lib.call("tise_stats_update_sum", self._h, _ptr(feats), feats.shape[0], feats.stride(0), _stream())

    def buffer(self):
        """The contiguous fp64 device buffer [S | s | n | pad] as a torch tensor VIEW (no copy);
        this is what the data-parallel driver all-reduces over RCCL."""
        p = ctypes.c_void_p()
        n = ctypes.c_size_t()
        _lib.call("tise_stats_buffer", self._h, ctypes.byref(p), ctypes.byref(n))
        return _wrap_device_doubles(p.value, n.value, self.device, owner=self)

    def count(self):
        return float(self.buffer()[self.dims * self.dims + self.dims].item())

    def finalize(self):
        """-> (mu (d,), sigma (d,d)) fp64 CUDA tensors; np.mean / np.cov(ddof=1) of everything fed."""
        mu = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        sigma = torch.empty((self.dims, self.dims), dtype=torch.float64, device=self.device)
        _lib.call("tise_stats_finalize", self._h, _ptr(mu), _ptr(sigma), _stream())
        return mu, sigma


def stats_update_grouped(accs, feats_sorted, offsets):
    """ONE launch: rows [offsets[g], offsets[g + 1]) of ``feats_sorted`` (fp32 CUDA (rows, d), sorted by group) are folded into
    ``accs[g]`` (StatsAccumulator) for every g (tise_stats_update_grouped)."""
    _require_cuda(feats_sorted)
    ng = len(accs)
    if ng == 0:
        return
    if feats_sorted.dim() != 2 or feats_sorted.dtype != torch.float32 or feats_sorted.shape[1] != accs[0].dims:
        raise ValueError("feats_sorted must be (rows, dims) float32")
    if feats_sorted.stride(1) != 1:
        feats_sorted = feats_sorted.contiguous()
    offs = [int(o) for o in offsets]
    if len(offs) != ng + 1 or offs[0] != 0 or offs[-1] != feats_sorted.shape[0]:
        raise ValueError("offsets must run from 0 to the row count, one more entry than groups")
    handles = (ctypes.c_void_p * ng)(*[a._h for a in accs])
    arr = (ctypes.c_int64 * (ng + 1))(*offs)
    _lib.call("tise_stats_update_grouped", handles, ng, _ptr(feats_sorted), arr, feats_sorted.stride(0), _stream())


def _wrap_device_doubles(ptr, n, device, owner=None):
    """Zero-copy torch view of `n` doubles at device address `ptr` (__cuda_array_interface__)."""
    class _Holder:
        pass
    holder = _Holder()
    holder.__cuda_array_interface__ = {
        "shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 3, "strides": None,
    }
    holder._owner = owner
    t = torch.as_tensor(holder, device=device)
    return t


class FrechetSolver:
    """Device Frechet distance (tise_frechet_* in include/tise_hip.h)."""

    def __init__(self, dims, device=None):
        self.dims = int(dims)
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.TiseLibraryError("FrechetSolver needs a HIP device")
        self._h = ctypes.c_void_p()
        self._pid = os.getpid()
        with torch.cuda.device(self.device):
            _lib.call("tise_frechet_create", self.dims, ctypes.byref(self._h))

    def close(self):
        if self._h and os.getpid() == getattr(self, "_pid", None):      # never from a forked child (StatsAccumulator.close)
            _lib.load().tise_frechet_destroy(self._h)
        self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _prep(self, t, shape):
        t = torch.as_tensor(t, dtype=torch.float64, device=self.device).contiguous()
        if tuple(t.shape) != shape:
            raise ValueError(f"expected shape {shape}, got {tuple(t.shape)}")
        return t

    def distance(self, mu1, sigma1, mu2, sigma2, diag_offset=0.0):
        """-> dict(fid, tr_covmean, diff2, tr1, tr2, rank, n_negative, flags); one device->host read."""
        d = self.dims
        mu1 = self._prep(mu1, (d,)); mu2 = self._prep(mu2, (d,))
        sigma1 = self._prep(sigma1, (d, d)); sigma2 = self._prep(sigma2, (d, d))
        out = torch.empty(_lib.TISE_FRECHET_OUT_DOUBLES, dtype=torch.float64, device=self.device)
        _lib.call("tise_frechet_distance", self._h, _ptr(mu1), _ptr(sigma1), _ptr(mu2), _ptr(sigma2),
                  float(diag_offset), _ptr(out), _stream())
        o = out.cpu().numpy()
        return {"fid": np.float64(o[0]), "tr_covmean": float(o[1]), "diff2": float(o[2]), "tr1": float(o[3]),
                "tr2": float(o[4]), "rank": int(o[5]), "n_negative": int(o[6]), "flags": int(o[7])}

    def prefactor(self, sigma, stream=None):
        """Start the pivoted Cholesky of ``sigma`` (d,d fp64 CUDA tensor, kept alive by the solver) on ``stream``
        (default: a private side stream, so it overlaps whatever the current stream is doing).  The host blocks
        until that stream has finished the factorisation (the numerical rank is read back), not the device."""
        sigma = self._prep(sigma, (self.dims, self.dims))
        if stream is None:
            if getattr(self, "_side", None) is None:
                self._side = torch.cuda.Stream(device=self.device)
            stream = self._side
        stream.wait_stream(torch.cuda.current_stream(self.device))        # sigma may still be in flight there
        self._pf_sigma = sigma
        _lib.call("tise_frechet_prefactor", self._h, _ptr(sigma), ctypes.c_void_p(stream.cuda_stream))
        sigma.record_stream(stream)
        return self

    def distance_prefactored(self, mu_f, mu_o, sigma_o):
        """Frechet distance between (mu_f, the sigma given to prefactor()) and (mu_o, sigma_o)."""
        d = self.dims
        if getattr(self, "_pf_sigma", None) is None:
            raise RuntimeError("prefactor() has not been called")
        mu_f = self._prep(mu_f, (d,)); mu_o = self._prep(mu_o, (d,))
        sigma_o = self._prep(sigma_o, (d, d))
        out = torch.empty(_lib.TISE_FRECHET_OUT_DOUBLES, dtype=torch.float64, device=self.device)
        _lib.call("tise_frechet_distance_prefactored", self._h, _ptr(mu_f), _ptr(self._pf_sigma), _ptr(mu_o), _ptr(sigma_o),
                  _ptr(out), _stream())
        o = out.cpu().numpy()
        return {"fid": np.float64(o[0]), "tr_covmean": float(o[1]), "diff2": float(o[2]), "tr1": float(o[3]),
                "tr2": float(o[4]), "rank": int(o[5]), "n_negative": int(o[6]), "flags": int(o[7])}

    def prefactor_ms(self):
        ms = ctypes.c_double()
        _lib.call("tise_frechet_prefactor_ms", self._h, ctypes.byref(ms))
        return ms.value

    def set_profiling(self, on=True):
        _lib.call("tise_frechet_set_profiling", self._h, 1 if on else 0)

    def phase_ms(self):
        """HIP-event phase times of the last distance() call: dict(pchol, gemm, sytrd, bisect, finish, rank)."""
        ms = (ctypes.c_double * 5)()
        r = ctypes.c_int()
        _lib.call("tise_frechet_phase_ms", self._h, ms, ctypes.byref(r))
        return {"pchol": ms[0], "gemm": ms[1], "sytrd": ms[2], "bisect": ms[3], "finish": ms[4], "rank": r.value}

    def eigvalsh(self, a):
        a = torch.as_tensor(a, dtype=torch.float64, device=self.device).contiguous()
        n = a.shape[0]
        w = torch.empty(n, dtype=torch.float64, device=self.device)
        _lib.call("tise_eigvalsh", self._h, _ptr(a), n, _ptr(w), _stream())
        return w

    def pivoted_cholesky(self, sigma):
        sigma = self._prep(sigma, (self.dims, self.dims))
        lt = torch.empty_like(sigma)
        r = ctypes.c_int()
        _lib.call("tise_pivoted_cholesky", self._h, _ptr(sigma), _ptr(lt), ctypes.byref(r), _stream())
        return lt, r.value


class InceptionScoreAccumulator:
    """Per-split additive IS* sums on the device (tise_is_update / tise_is_finalize)."""

    RULES = {"coco": 0, "bird": 0, "ois": 1}

    def __init__(self, num_classes, n_total, temperature, splits=10, rule="coco", drop_first_class=False, device=None):
        self.device = torch.device(device if device is not None else "cuda")
        if self.device.type != "cuda":
            raise _lib.TiseLibraryError("InceptionScoreAccumulator needs a HIP device")
        self.C = int(num_classes)
        self.drop = 1 if drop_first_class else 0
        self.Ce = self.C - self.drop
        self.n_total = int(n_total)
        self.T = float(temperature)
        self.splits = int(splits)
        self.rule = self.RULES[rule] if isinstance(rule, str) else int(rule)
        self.acc = torch.zeros(self.splits * (1 + self.Ce), dtype=torch.float64, device=self.device)
        self._ws = None

    def update(self, logits, idx_base):
        """logits: (rows, C) fp32 CUDA tensor whose rows have global indices idx_base..idx_base+rows-1."""
        _require_cuda(logits)
        if logits.dim() != 2 or logits.shape[1] != self.C or logits.dtype != torch.float32:
            raise ValueError(f"logits must be (rows,{self.C}) float32")
        if logits.stride(1) != 1:
            logits = logits.contiguous()
        rows = logits.shape[0]
        if self._ws is None or self._ws.numel() < 2 * rows:
            self._ws = torch.empty(2 * max(rows, 1024), dtype=torch.float64, device=self.device)
        _lib.call("tise_is_update", _ptr(logits), rows, logits.stride(0), self.C, self.T, self.drop, int(idx_base),
                  self.n_total, self.splits, self.rule, _ptr(self.acc), _ptr(self._ws), _stream())

    def finalize(self):
        """-> (mean, std, scores[splits]) as Python floats / numpy."""
        out = torch.empty(2 + self.splits, dtype=torch.float64, device=self.device)
        _lib.call("tise_is_finalize", _ptr(self.acc), self.Ce, self.n_total, self.splits, self.rule, _ptr(out), _stream())
        o = out.cpu().numpy()
        return float(o[0]), float(o[1]), o[2:].copy()


def gemm_f64(a, b):
    """C = A @ B in fp64 through the MFMA tile kernel (any strides); test/bench helper."""
    _require_cuda(a, b)
    m, k = a.shape
    k2, n = b.shape
    assert k == k2 and a.dtype == torch.float64 and b.dtype == torch.float64
    c = torch.empty((m, n), dtype=torch.float64, device=a.device)
    _lib.call("tise_gemm_f64", _ptr(a), a.stride(0), a.stride(1), _ptr(b), b.stride(0), b.stride(1), _ptr(c), n,
              m, n, k, _stream())
    return c


def cosine_top1(img_emb, txt_emb, txt_index=None, normalize=True, logit_scale=100.0, want_p0=True):
    """Top-1 text retrieval (RP_coco.py:72-78 / PA.py:37-42) for all items at once.

    img_emb (n, d), txt_emb (rows, d): fp32 or fp16 device tensors; txt_index (n, c) int32 rows of txt_emb per item,
    candidate 0 = true caption (None: txt_emb is (n*c, d), item-major).  Returns (top1 int32 (n,), p0 fp32 (n,) or None).
    """
    _require_cuda(img_emb, txt_emb, txt_index)
    assert img_emb.dim() == 2 and txt_emb.dim() == 2 and img_emb.shape[1] == txt_emb.shape[1]
    assert img_emb.dtype == txt_emb.dtype and img_emb.dtype in (torch.float32, torch.float16)
    img_emb, txt_emb = img_emb.contiguous(), txt_emb.contiguous()
    n, d = img_emb.shape
    if txt_index is not None:
        assert txt_index.dtype == torch.int32 and txt_index.dim() == 2 and txt_index.shape[0] == n
        txt_index = txt_index.contiguous()
        c = txt_index.shape[1]
    else:
        assert txt_emb.shape[0] % max(n, 1) == 0
        c = txt_emb.shape[0] // max(n, 1)
    top1 = torch.empty(n, dtype=torch.int32, device=img_emb.device)
    p0 = torch.empty(n, dtype=torch.float32, device=img_emb.device) if want_p0 else None
    _lib.call("tise_cosine_top1", _ptr(img_emb), _ptr(txt_emb), _ptr(txt_index) if txt_index is not None else None,
              ctypes.c_int64(n), int(c), int(d), 0 if img_emb.dtype == torch.float32 else 1, 1 if normalize else 0,
              ctypes.c_float(logit_scale), _ptr(top1), _ptr(p0) if p0 is not None else None, _stream())
    return top1, p0
