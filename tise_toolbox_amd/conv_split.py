"""Host side of the split-precision fp16-MFMA convolution (csrc/conv_split.hip).

A value v is carried as two fp16 numbers, v ~= hi + lo * 2**-11.  A split tensor of C channels is an fp16 tensor
(..., 2*C): NHWC, inside a pixel the channels in blocks of 32 with the halves side by side -- [hi x32 | lo x32] per
block, one 128-byte line = one K-step of the convolution -- and a last block [hi x16 | lo x16] when C % 32 == 16
(csrc/common.h).  ``split`` / ``merge`` convert between fp32 (..., C) and that form, ``new_split`` allocates one;
``SplitConv`` packs BatchNorm-folded conv weights once
(power-of-two per-channel scaling so the fp16 halves stay normal, K = (kh, kw, cin) with cin fastest,
zero padding to the kernel's tile multiples) and launches ``tise_conv_split_f16``.
"""
import ctypes
import os

import torch

from . import _lib

LO_SCALE = 2048.0   # 2**11


class ConvSeg(ctypes.Structure):
    _fields_ = [("c0", ctypes.c_int), ("c1", ctypes.c_int), ("dst", ctypes.c_void_p), ("ld", ctypes.c_longlong),
                ("off", ctypes.c_int), ("mode", ctypes.c_int)]


class ConvArgs(ctypes.Structure):
    _fields_ = [("x", ctypes.c_void_p), ("w", ctypes.c_void_p),
                ("w_plane", ctypes.c_longlong), ("scale", ctypes.c_void_p), ("bias", ctypes.c_void_p),
                ("N", ctypes.c_int), ("H", ctypes.c_int), ("W", ctypes.c_int), ("Cin", ctypes.c_int),
                ("KH", ctypes.c_int), ("KW", ctypes.c_int), ("SH", ctypes.c_int), ("SW", ctypes.c_int),
                ("PH", ctypes.c_int), ("PW", ctypes.c_int), ("OH", ctypes.c_int), ("OW", ctypes.c_int),
                ("Cout", ctypes.c_int), ("K", ctypes.c_int), ("Kpad", ctypes.c_int), ("M", ctypes.c_longlong),
                ("nseg", ctypes.c_int), ("seg", ConvSeg * 4),
                ("out_hp", ctypes.c_int), ("out_wp", ctypes.c_int), ("out_y0", ctypes.c_int), ("out_x0", ctypes.c_int)]


def split_planes(x):
    """fp32 tensor -> (2, *x.shape) fp16: plane 0 = hi, plane 1 = (x - hi) * 2**11 (the weight format of the generic kernel)."""
    hi = x.half()
    lo = ((x - hi.float()) * LO_SCALE).half()
    return torch.stack([hi, lo], 0)


def _interleave(hi, lo):
    """(..., C) hi / lo -> (..., 2C) in the layout of csrc/common.h."""
    C = hi.shape[-1]
    assert C % 16 == 0, "split tensors need C % 16 == 0"
    full = C & ~31
    lead = hi.shape[:-1]
    parts = []
    if full:
        blk = torch.stack([hi[..., :full].reshape(*lead, full // 32, 32), lo[..., :full].reshape(*lead, full // 32, 32)], -2)
        parts.append(blk.reshape(*lead, 2 * full))
    if C > full:
        parts.append(torch.cat([hi[..., full:], lo[..., full:]], -1))
    return torch.cat(parts, -1).contiguous() if len(parts) > 1 else parts[0].contiguous()


def split(x):
    """fp32 (..., C) -> split tensor (..., 2C) fp16."""
    hi = x.half()
    lo = ((x - hi.float()) * LO_SCALE).half()
    return _interleave(hi, lo)


def halves(t):
    """split tensor (..., 2C) -> (hi, lo), each (..., C) fp16."""
    C = t.shape[-1] // 2
    full = C & ~31
    lead = t.shape[:-1]
    his, los = [], []
    if full:
        blk = t[..., :2 * full].reshape(*lead, full // 32, 2, 32)
        his.append(blk[..., 0, :].reshape(*lead, full))
        los.append(blk[..., 1, :].reshape(*lead, full))
    if C > full:
        his.append(t[..., 2 * full:2 * full + (C - full)])
        los.append(t[..., 2 * full + (C - full):])
    return torch.cat(his, -1), torch.cat(los, -1)


def merge(t):
    """split tensor (..., 2C) -> fp32 (..., C)."""
    hi, lo = halves(t)
    return hi.float() + lo.float() * (1.0 / LO_SCALE)


def new_split(n, h, w, c, device):
    """Uninitialised split tensor of c channels."""
    assert c % 16 == 0
    return torch.empty((n, h, w, 2 * c), dtype=torch.float16, device=device)


# relative efficiency of the kernel by tile width (tools/conv_split_probe.py): narrow tiles re-read the
# pixel operand more often per MFMA
_TN_EFF = {1: 0.55, 2: 0.78, 3: 0.90, 4: 1.0, 5: 1.0}


def pick_tn(cout):
    """Tile width 32*tn (tn in 1..5): least padded work weighted by the measured tile efficiency.  (Timing the
    candidate widths per layer and input shape at run time -- every width gives the same bits -- picked differently on
    a quarter of the layers and was 2 % faster summed over isolated launches, tools/tn_sweep.py, but 21.10-21.14 k
    vs 21.10 k images/s in the bench: not kept.)"""
    best = None
    for tn in (5, 4, 3, 2, 1):
        bn = 32 * tn
        padded = -(-cout // bn) * bn
        cost = padded / _TN_EFF[tn]
        if best is None or cost < best[0]:
            best = (cost, tn)
    return best[1]


def rowwin_applies(cin, cout, kh, kw, stride, padding, tn):
    """Layers the row-window kernel serves: stride 1, a filter wider than one pixel, tile widths 2..4 (its LDS budget:
    two window buffers + two weight stages, two workgroups per CU)."""
    return tuple(stride) == (1, 1) and 2 <= kw <= 8 and cin % 16 == 0 and cin >= 32 and tn in (2, 3, 4)


def rowwin_fits(ow, kw):
    """The row-window kernel's window holds 5 or 6 pieces of 8 rows per wave (csrc/conv_split.hip, rowwin_np)."""
    j = (ow + 126) // ow + 1
    rows = 128 + j * (kw - 1) + 1
    return -(-rows // 32) in (5, 6)


# conv_pipe.hip: configuration 34 = register-resident-weights sliding-window kernel (Cin = 32, 3x3, stride 1), up to 64 couts per launch
PIPE_BN = {34: 64}


def pipe_fits(w, cout):
    """LDS budget of configuration 34 (conv_pipe.hip launch_regw32: input ring + epilogue area + staging <= 160 KB)."""
    tpi = 64 // cout
    r16 = (128 * tpi + 2 * w + 2 + 15) & ~15
    pf = 2 if cout == 64 else 1
    return w >= 8 and (r16 + 128 * tpi * pf) * 128 + 2048 + 8 * 32 * (128 + 16) <= 160 * 1024


def pool_output_fits(w, ow):
    """Configuration 34 with the max-pool in its epilogue (conv_pipe.hip launch_regw32_pool): a 128-pixel tile spans at most
    two grid rows, the horizontal buffer keeps 130 columns, ring + column state within 160 KB of LDS."""
    r16 = (128 + 2 * w + 2 + 15) & ~15
    return 128 <= w <= 260 and ow >= 3 and (r16 + 256) * 128 + 2048 + ow * 256 + 130 * 256 <= 160 * 1024


class SplitConv:
    """One (possibly channel-concatenated) convolution with folded scale/bias, packed for the kernel."""

    # measurement hook (bench.py): when a list, every launch appends (start_event, end_event, flop)
    timer = None

    def __init__(self, weight, bias, stride, padding, device, tn=None, variant=None, pipe_cfg=None, korder=None):
        """weight: (Cout, Cin, KH, KW) fp32 (BatchNorm already folded), bias: (Cout,) fp32."""
        cout, cin, kh, kw = weight.shape
        assert cin % 16 == 0 and cin >= 32, "conv_split needs Cin % 16 == 0 and Cin >= 32"
        self.cout, self.cin, self.kh, self.kw = cout, cin, kh, kw
        self.stride = tuple(stride)
        self.padding = tuple(padding)
        self.tn = tn or pick_tn(cout)
        self.pipe_cfg = None
        self._fallback, self._orig = None, None
        # kernel variant: "fast" = LDS-DMA staging with hoisted addressing (default); "glds" = its generic form
        # (addresses recomputed per K-step, natural K order, any M: the reference kernel of the tests);
        # "pipe" = conv_pipe.hip configuration 34 (Conv2d_2a, Conv2d_2b)
        # "rowwin" = row-window kernel (the kw taps of a filter row share one fetch of the pixel operand); "auto" (the
        # default) = rowwin where it applies and measured faster (tools/conv_rowwin_probe.py), else fast
        self.variant = variant or os.environ.get("TISE_CONV_VARIANT", "auto")
        was_auto = self.variant == "auto"
        if self.variant == "auto":
            self.variant = "rowwin" if rowwin_applies(cin, cout, kh, kw, self.stride, self.padding, self.tn) else "fast"
        # K order of the default kernel: "tap" = (tap, 32-channel block), the generic kernel's order (bit-identical to it);
        # "block" = (block, tap) for unpadded multi-tap layers with Cin % 32 == 0 -- the stride-2 3x3 layers of Mixed_6a / 7a --
        # whose tap-shifted re-reads then hit L2 (csrc/conv_split.hip CBT; chosen when the variant is left to "auto",
        # TISE_CONV_KORDER=tap keeps the tap-major order everywhere)
        can_block = (self.variant == "fast" and self.padding == (0, 0) and kh * kw > 1 and cin % 32 == 0 and self.tn >= 2)
        if korder is None:
            korder = "block" if (was_auto and can_block and os.environ.get("TISE_CONV_KORDER", "block") == "block") else "tap"
        if korder not in ("tap", "block") or (korder == "block" and not can_block):
            raise ValueError(f"korder {korder!r} does not apply to this layer")
        self.korder = korder
        if self.variant not in ("fast", "glds", "pipe", "rowwin"):
            raise ValueError(f"unknown conv variant {self.variant!r} (round 2 removed reg / glds3 / gldsb / win / spec)")
        # rows of zero weights / scale / bias up to the widest tile grid any tile width may use
        self.cout_pad = max(-(-cout // (32 * t)) * 32 * t for t in (1, 2, 3, 4, 5))
        if self.variant == "pipe":                              # resident-weights sliding-window kernel (conv_pipe.hip)
            self.pipe_cfg = 34 if pipe_cfg is None else pipe_cfg
            self._orig = (weight.detach().float(), bias.detach().float())                   # see __call__
        self.k = kh * kw * cin
        self.kpad = -(-self.k // 32) * 32
        # The packing runs ON THE DEVICE (round 6: 68 layers of host tensor arithmetic were 0.12 s of a CLI's start-up).  Every
        # step is exact -- scaling by powers of two, permutations, zero padding, round-to-nearest-even conversions to fp16 --
        # so host and device give the same bits; the power of two comes from frexp (the exponent field), not from log2.
        dev = torch.device(device)
        w = weight.detach().float().to(dev)
        amax = w.abs().reshape(cout, -1).amax(1).clamp_min(1e-30)
        _, ex = torch.frexp(amax)                                   # amax = m * 2**ex, m in [0.5, 1)
        pre = torch.ldexp(torch.ones_like(amax), 1 - ex)            # exact powers of two: max |w * pre| in [1, 2)
        wk = (w * pre.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(cout, self.k)
        wp = torch.zeros((self.cout_pad, self.kpad), dtype=torch.float32, device=dev)
        wp[:cout, :self.k] = wk
        self.w = split_planes(wp).contiguous()                      # (2, Cout_pad, Kpad) fp16
        if self.variant == "fast" and cin % 32 == 16:
            # default kernel, Cin = 32 * nfull + 16: K order = (tap, full 32-channel block) for all taps, then the
            # 16-channel tails two taps per 32-wide step (conv_split.hip, conv_split_fast_kernel)
            ntaps, nfull = kh * kw, cin // 32
            w3 = torch.zeros((self.cout_pad, ntaps, cin), dtype=torch.float32, device=dev)
            w3[:cout] = wk.reshape(cout, ntaps, cin)
            full = w3[:, :, :nfull * 32].reshape(self.cout_pad, ntaps * nfull * 32)
            tails = torch.zeros((self.cout_pad, (ntaps + 1) // 2 * 2, 16), dtype=torch.float32, device=dev)
            tails[:, :ntaps] = w3[:, :, nfull * 32:]
            self.w = split_planes(torch.cat([full, tails.reshape(self.cout_pad, -1)], 1)).contiguous()
            self.kpad = self.w.shape[2]
        if self.korder == "block":
            ntaps, nblk = kh * kw, cin // 32
            w4 = torch.zeros((self.cout_pad, ntaps, nblk, 32), dtype=torch.float32, device=dev)
            w4[:cout] = wk.reshape(cout, ntaps, nblk, 32)
            self.w = split_planes(w4.permute(0, 2, 1, 3).reshape(self.cout_pad, -1)).contiguous()
        # the default kernel reads the weights as ONE 128-byte line per (cout, 32-wide K block): [hi 32 | lo 32]
        # (conv_split_fast_kernel: 128-byte LDS-DMA rows); the planar (2, Cout_pad, Kpad) form serves "glds" / "pipe"
        self.w_fast = None
        if self.variant == "rowwin":
            self._orig = (weight.detach().float(), bias.detach().float())                   # see __call__
            # row-window kernel: K order (kh, 32-channel block, kw) -- the kw taps of a (kh, block) group share one window;
            # Cin = 32 n + 16: each kh ends with the 16-channel tails, two taps per 32-wide step
            assert cin % 16 == 0 and self.tn in (2, 3, 4)
            nfull, tail = cin // 32, cin % 32
            w4 = torch.zeros((self.cout_pad, kh, kw, cin), dtype=torch.float32, device=dev)
            w4[:cout] = wk.reshape(cout, kh, kw, cin)
            per_kh = []
            for a in range(kh):
                if nfull:
                    full = w4[:, a, :, :nfull * 32].reshape(self.cout_pad, kw, nfull, 32).permute(0, 2, 1, 3)
                    per_kh.append(full.reshape(self.cout_pad, nfull * kw * 32))
                if tail:
                    tl = torch.zeros((self.cout_pad, (kw + 1) // 2 * 2, 16), dtype=torch.float32, device=dev)
                    tl[:, :kw] = w4[:, a, :, nfull * 32:]
                    per_kh.append(tl.reshape(self.cout_pad, -1))
            self.w = split_planes(torch.cat(per_kh, 1)).contiguous()
            self.kpad = self.w.shape[2]
        if self.variant in ("fast", "rowwin"):
            hi, lo = self.w[0], self.w[1]
            self.w_fast = torch.stack([hi.reshape(self.cout_pad, -1, 32), lo.reshape(self.cout_pad, -1, 32)], 2).contiguous()
            self.w = None
        sc = torch.zeros(self.cout_pad, dtype=torch.float32, device=dev)
        sc[:cout] = torch.ldexp(torch.ones_like(amax), ex - 1)      # 1 / pre, exactly
        bs = torch.zeros(self.cout_pad, dtype=torch.float32, device=dev)
        bs[:cout] = bias.detach().float().to(dev)
        self.scale = sc
        self.bias = bs

    def out_hw(self, h, w):
        oh = (h + 2 * self.padding[0] - self.kh) // self.stride[0] + 1
        ow = (w + 2 * self.padding[1] - self.kw) // self.stride[1] + 1
        return oh, ow

    def pooled_out_hw(self, h, w):
        """Output grid of ``__call__(..., pooled_input=True)``: max_pool2d(3, stride 2) of the input, then this 1x1 conv."""
        return (h - 3) // 2 + 1, (w - 3) // 2 + 1

    def __call__(self, x, segs, pooled_input=False, out_pad=None, pool_output=False, pool_h=False):
        """x: split tensor (N, H, W, 2*Cin) fp16.  segs: list of (c0, c1, dst_tensor, dst_off, mode):
        mode 0 -> dst is a split tensor (N, OH, OW, 2*C), mode 1 -> dst is a (N, OH, OW, C) fp32 tensor.
        ``pooled_input``: the convolution (1x1, Cin % 32 == 0, default packing) reads max_pool2d(x, 3, stride 2) -- the
        pool is taken while loading the operand (conv_poolin_kernel), bit-identical to pooling first.
        ``out_pad`` = (hp, wp, y0, x0) (sliding-window kernels only): the destination tensors are (N, hp, wp, ...) images
        and the (OH, OW) result is written at offset (y0, x0) inside them (the rest is left untouched).
        ``pool_output`` (configuration 34, 64 couts, unpadded: Conv2d_2b on its zero-bordered input): the destinations are split
        tensors of max_pool2d(result, 3, stride 2) -- the pool is taken in the kernel's epilogue, bit-identical to
        pooling the stored result; returns the pooled (OH', OW').
        ``pool_h`` (row-window kernel, tile width 3: Conv2d_4a): the destinations are split tensors of the HORIZONTAL half of
        max_pool2d(result, 3, stride 2) -- (N, OH, (OW - 3) // 2 + 1, ...) -- taken in the epilogue; the consumer finishes the
        pool with ``pooled_input="v"`` (three vertical taps).  Together bit-identical to pooling the stored result."""
        assert x.dtype == torch.float16 and x.dim() == 4 and x.shape[3] == 2 * self.cin and x.is_contiguous()
        n, h, w, _ = x.shape
        oh, ow = self.out_hw(h, w)
        if pooled_input:
            assert (self.kh, self.kw, self.stride, self.padding) == (1, 1, (1, 1), (0, 0)) and self.cin % 32 == 0
            assert self.variant == "fast" and self.w_fast is not None and h >= 3 and (w >= 3 or pooled_input == "v")
            # the kernel's cout tiles: one of 128 couts, else tiles of 256 (weights / scale / bias are zero-padded to cout_pad rows)
            assert self.cout_pad >= (128 if self.cout <= 128 else -(-self.cout // 256) * 256)
            oh, ow = self.pooled_out_hw(h, w)
            if pooled_input == "v":                             # the producer took the horizontal half (pool_h)
                ow = w
        if self.variant == "rowwin" and not rowwin_fits(ow, self.kw):
            # rows so short that the window of a 128-pixel tile needs more than the kernel's six pieces per wave (OW < 7
            # at KW = 3): the default kernel serves the layer from its own packing, built on first use
            if self._fallback is None:
                self._fallback = SplitConv(self._orig[0], self._orig[1], self.stride, self.padding, x.device, tn=self.tn, variant="fast")
            return self._fallback(x, segs)
        if self.pipe_cfg is not None and out_pad is None and not pipe_fits(w, self.cout):
            # image rows so long that the sliding ring does not fit the LDS (W > ~230): the default kernel, built on first use
            if self._fallback is None:
                self._fallback = SplitConv(self._orig[0], self._orig[1], self.stride, self.padding, x.device, variant="fast")
            return self._fallback(x, segs)
        a = ConvArgs()
        a.x = x.data_ptr()
        wt = self.w_fast if self.w_fast is not None else self.w
        a.w = wt.data_ptr(); a.w_plane = 0 if self.w is None else self.w.stride(0)
        a.scale = self.scale.data_ptr(); a.bias = self.bias.data_ptr()
        a.N, a.H, a.W, a.Cin = n, h, w, self.cin
        a.KH, a.KW, a.SH, a.SW, a.PH, a.PW = self.kh, self.kw, self.stride[0], self.stride[1], self.padding[0], self.padding[1]
        a.OH, a.OW = oh, ow
        a.Cout, a.K, a.Kpad = self.cout, self.k, self.kpad
        a.M = n * oh * ow
        a.nseg = len(segs) | getattr(self, "debug_flags", 0)
        dshape = (n, oh, ow)
        if pool_h:
            assert self.variant == "rowwin" and self.tn == 3 and rowwin_fits(ow, self.kw) and ow >= 3 and out_pad is None
            assert not pooled_input and not pool_output and all(sg[4] == 0 for sg in segs)
            dshape = (n, oh, (ow - 3) // 2 + 1)
        if pool_output:
            assert self.pipe_cfg == 34 and self.cout == 64 and self.padding == (0, 0) and out_pad is None and not pooled_input
            assert pool_output_fits(w, ow) and oh >= 3 and all(sg[4] == 0 for sg in segs)
            dshape = (n, (oh - 3) // 2 + 1, (ow - 3) // 2 + 1)
        if out_pad is not None:
            hp, wp, y0, x0 = out_pad
            assert self.pipe_cfg is not None and not pooled_input and y0 + oh <= hp and x0 + ow <= wp and min(y0, x0) >= 0
            a.out_hp, a.out_wp, a.out_y0, a.out_x0 = hp, wp, y0, x0
            dshape = (n, hp, wp)
        for i, (c0, c1, dst, off, mode) in enumerate(segs):
            s = a.seg[i]
            s.c0, s.c1, s.off, s.mode = c0, c1, off, mode
            s.dst = dst.data_ptr()
            if mode == 0:
                assert dst.dtype == torch.float16 and dst.shape[:3] == dshape and dst.is_contiguous()
                s.ld = dst.shape[3] // 2
            else:
                assert dst.dtype == torch.float32 and dst.shape[:3] == dshape and dst.is_contiguous()
                s.ld = dst.shape[3]
        if getattr(self, "debug_ptr", None):                    # tools/conv_stamps.py
            a.seg[3].dst = self.debug_ptr
        stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        tn = self.tn
        timer = SplitConv.timer
        if timer is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        if pooled_input:
            code = min(max(tn, 2), 4) | 256 | (2048 if pooled_input == "v" else 0)
        elif self.pipe_cfg is not None:
            code = 512 | self.pipe_cfg | (1024 if pool_output else 0)
        else:
            code = tn | {"glds": 16, "fast": 128, "rowwin": 64}[self.variant] | (2048 if pool_h else 0) | (4096 if self.korder == "block" else 0)
        _lib.call("tise_conv_split_f16", ctypes.byref(a), code, stream)
        if timer is not None:
            e1.record()
            timer.append((e0, e1, 2.0 * a.M * self.cout * self.k))
        return (dshape[1], dshape[2]) if (pool_output or pool_h) else (oh, ow)
