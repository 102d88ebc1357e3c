"""Fused InceptionV3 trunk executor: MIOpen convolutions + hand-written HIP epilogues.

Same function as ``InceptionV3.forward`` (mirror of ``image_realism/FID/inception.py:100-134``) for
inputs that are already 299x299 and affine-normalised by the resize kernel, restructured for MI355X
after the first rocprof pass (``profiles/r01a_*``: 35 % of a step was not convolution):

* eval-mode BatchNorm folded into the weights; the bias + ReLU of every conv is ONE in-place pass
  (``tise_bias_relu_nhwc``) instead of PyTorch's separate bias-add and ReLU kernels;
* the 1x1 convolutions that read the same block input (torchvision InceptionA/C/D/E: branch1x1,
  the ``*_1`` reducers and the pool branch's 1x1) are one convolution with concatenated output
  channels (94 -> 66 conv launches, larger GEMM N);
* the pool branch is evaluated as 1x1 conv THEN 3x3 average (both linear, they commute, border
  included) so the pool touches 32..192 channels instead of 192..2048, fused with bias + ReLU
  (``tise_avgpool3_bias_relu_nhwc``);
* every branch's last epilogue writes straight into its channel slice of the block output -- no
  ``torch.cat``;
* the two stem max-pools absorb the preceding conv's bias + ReLU (``tise_maxpool3s2_nhwc``).

Numerics: fp32 throughout; relative to the reference graph order only fp32 summation order changes
(pool/conv commutation, fused-1x1 kernel choice).  ``tests/test_gpu_pipeline.py`` checks it against the
unfused module and against the CPU oracle.
"""
import ctypes
import os

import torch

from . import _lib
from .inception import BasicConv2d, InceptionV3


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


class _Conv:
    """Folded conv parameters on the device: weight (Cout,Cin,kh,kw) channels-last, bias (Cout,)."""

    __slots__ = ("w", "b", "stride", "padding", "cout")

    def __init__(self, mods, device):
        ws, bs = [], []
        for m in mods:
            bn = m.bn
            scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
            ws.append(m.conv.weight * scale.view(-1, 1, 1, 1))
            bs.append(bn.bias - bn.running_mean * scale)
        m0 = mods[0].conv
        for m in mods[1:]:
            assert m.conv.kernel_size == m0.kernel_size and m.conv.stride == m0.stride and m.conv.padding == m0.padding
        self.w = torch.cat(ws, 0).to(device).contiguous(memory_format=torch.channels_last)
        self.b = torch.cat(bs, 0).to(device).contiguous()
        self.stride, self.padding = m0.stride, m0.padding
        self.cout = self.w.shape[0]


class FusedTrunk:
    """Callable: (N,299,299,3)-storage channels-last input -> pool3 features (N, 2048) fp32."""

    def __init__(self, model, device):
        assert isinstance(model, InceptionV3)
        self.device = torch.device(device)
        with torch.no_grad():
            mods = dict(model.named_modules())

            def conv(*names):
                ms = [mods[n] for n in names]
                assert all(isinstance(m, BasicConv2d) for m in ms)
                return _Conv(ms, self.device)

            b0, b1, b2, b3 = "blocks.0.", "blocks.1.", "blocks.2.", "blocks.3."
            self.last_block = model.last_needed_block
            self.c1a, self.c2a, self.c2b = conv(b0 + "0"), conv(b0 + "1"), conv(b0 + "2")
            if self.last_block >= 1:
                self.c3b, self.c4a = conv(b1 + "0"), conv(b1 + "1")
            self.blocks = []
            if self.last_block >= 2:
                for i, kind in enumerate("AAABCCCC"):
                    self.blocks.append((kind, self._block_params(kind, b2 + str(i) + ".", conv)))
            if self.last_block >= 3:
                for i, kind in enumerate("DEE"):
                    self.blocks.append((kind, self._block_params(kind, b3 + str(i) + ".", conv)))

    @staticmethod
    def _block_params(kind, p, conv):
        if kind == "A":
            return dict(f=conv(p + "branch1x1", p + "branch5x5_1", p + "branch3x3dbl_1", p + "branch_pool"),
                        c5=conv(p + "branch5x5_2"), d2=conv(p + "branch3x3dbl_2"), d3=conv(p + "branch3x3dbl_3"))
        if kind == "B":
            return dict(c3=conv(p + "branch3x3"), d1=conv(p + "branch3x3dbl_1"), d2=conv(p + "branch3x3dbl_2"),
                        d3=conv(p + "branch3x3dbl_3"))
        if kind == "C":
            return dict(f=conv(p + "branch1x1", p + "branch7x7_1", p + "branch7x7dbl_1", p + "branch_pool"),
                        s2=conv(p + "branch7x7_2"), s3=conv(p + "branch7x7_3"), d2=conv(p + "branch7x7dbl_2"),
                        d3=conv(p + "branch7x7dbl_3"), d4=conv(p + "branch7x7dbl_4"), d5=conv(p + "branch7x7dbl_5"))
        if kind == "D":
            return dict(f=conv(p + "branch3x3_1", p + "branch7x7x3_1"), c32=conv(p + "branch3x3_2"),
                        s2=conv(p + "branch7x7x3_2"), s3=conv(p + "branch7x7x3_3"), s4=conv(p + "branch7x7x3_4"))
        if kind == "E":
            return dict(f=conv(p + "branch1x1", p + "branch3x3_1", p + "branch3x3dbl_1", p + "branch_pool"),
                        a2=conv(p + "branch3x3_2a"), b2=conv(p + "branch3x3_2b"), d2=conv(p + "branch3x3dbl_2"),
                        a3=conv(p + "branch3x3dbl_3a"), b3=conv(p + "branch3x3dbl_3b"))
        raise ValueError(kind)

    # ---- primitive ops on NHWC tensors ---------------------------------------------------------------
    @staticmethod
    def _conv(x, c):
        """x: (N,H,W,Cin) contiguous -> raw conv output (N,OH,OW,Cout) contiguous (no bias, no ReLU)."""
        y = torch.conv2d(x.permute(0, 3, 1, 2), c.w, None, c.stride, c.padding).permute(0, 2, 3, 1)
        return y if y.is_contiguous() else y.contiguous()

    @staticmethod
    def _bias_relu(raw, bias, x_off=0, C=None, out=None, out_off=0):
        """max(raw[..., x_off:x_off+C] + bias, 0) -> out[..., out_off:out_off+C] (in place when out is None
        and the slice is the whole tensor; a new packed tensor when out is None and it is a sub-slice)."""
        n, h, w, ld = raw.shape
        C = ld if C is None else C
        if out is None:
            out = raw if (x_off == 0 and C == ld) else torch.empty((n, h, w, C), dtype=raw.dtype, device=raw.device)
        _lib.call("tise_bias_relu_nhwc", _p(raw), ld, x_off, n * h * w, C, _p(bias), _p(out), out.shape[3], out_off,
                  _stream())
        return out

    @staticmethod
    def _avgpool_bias_relu(raw, bias, x_off, C, out, out_off):
        n, h, w, ld = raw.shape
        _lib.call("tise_avgpool3_bias_relu_nhwc", _p(raw), ld, x_off, n, h, w, C, _p(bias), _p(out), out.shape[3],
                  out_off, _stream())

    @staticmethod
    def _maxpool(x, bias=None, out=None, out_off=0):
        n, h, w, C = x.shape
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        if out is None:
            out = torch.empty((n, oh, ow, C), dtype=x.dtype, device=x.device)
        _lib.call("tise_maxpool3s2_nhwc", _p(x), C, 0, n, h, w, C, _p(bias) if bias is not None else None, _p(out),
                  out.shape[3], out_off, _stream())
        return out

    # ---- blocks -------------------------------------------------------------------------------------
    def _block_a(self, x, P):
        n, h, w, _ = x.shape
        f, c5, d2, d3 = P["f"], P["c5"], P["d2"], P["d3"]
        pf = f.cout - 176
        out = torch.empty((n, h, w, 224 + pf), dtype=x.dtype, device=x.device)
        raw = self._conv(x, f)                                       # [1x1:64 | 5x5_1:48 | dbl_1:64 | pool:pf]
        self._bias_relu(raw, f.b[0:64], 0, 64, out, 0)
        t5 = self._bias_relu(raw, f.b[64:112], 64, 48)
        t3 = self._bias_relu(raw, f.b[112:176], 112, 64)
        self._avgpool_bias_relu(raw, f.b[176:], 176, pf, out, 224)
        self._bias_relu(self._conv(t5, c5), c5.b, 0, 64, out, 64)
        t3 = self._bias_relu(self._conv(t3, d2), d2.b)
        self._bias_relu(self._conv(t3, d3), d3.b, 0, 96, out, 128)
        return out

    def _block_b(self, x, P):
        n, h, w, cin = x.shape
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        out = torch.empty((n, oh, ow, 384 + 96 + cin), dtype=x.dtype, device=x.device)
        self._bias_relu(self._conv(x, P["c3"]), P["c3"].b, 0, 384, out, 0)
        t = self._bias_relu(self._conv(x, P["d1"]), P["d1"].b)
        t = self._bias_relu(self._conv(t, P["d2"]), P["d2"].b)
        self._bias_relu(self._conv(t, P["d3"]), P["d3"].b, 0, 96, out, 384)
        self._maxpool(x, None, out, 480)
        return out

    def _block_c(self, x, P):
        n, h, w, _ = x.shape
        f = P["f"]
        c7 = (f.cout - 384) // 2
        out = torch.empty((n, h, w, 768), dtype=x.dtype, device=x.device)
        raw = self._conv(x, f)                                       # [1x1:192 | 7x7_1:c7 | dbl_1:c7 | pool:192]
        self._bias_relu(raw, f.b[0:192], 0, 192, out, 0)
        t7 = self._bias_relu(raw, f.b[192:192 + c7], 192, c7)
        td = self._bias_relu(raw, f.b[192 + c7:192 + 2 * c7], 192 + c7, c7)
        self._avgpool_bias_relu(raw, f.b[192 + 2 * c7:], 192 + 2 * c7, 192, out, 576)
        t7 = self._bias_relu(self._conv(t7, P["s2"]), P["s2"].b)
        self._bias_relu(self._conv(t7, P["s3"]), P["s3"].b, 0, 192, out, 192)
        for k in ("d2", "d3", "d4"):
            td = self._bias_relu(self._conv(td, P[k]), P[k].b)
        self._bias_relu(self._conv(td, P["d5"]), P["d5"].b, 0, 192, out, 384)
        return out

    def _block_d(self, x, P):
        n, h, w, cin = x.shape
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        f = P["f"]
        out = torch.empty((n, oh, ow, 320 + 192 + cin), dtype=x.dtype, device=x.device)
        raw = self._conv(x, f)                                       # [3x3_1:192 | 7x7x3_1:192]
        t3 = self._bias_relu(raw, f.b[0:192], 0, 192)
        t7 = self._bias_relu(raw, f.b[192:384], 192, 192)
        self._bias_relu(self._conv(t3, P["c32"]), P["c32"].b, 0, 320, out, 0)
        for k in ("s2", "s3"):
            t7 = self._bias_relu(self._conv(t7, P[k]), P[k].b)
        self._bias_relu(self._conv(t7, P["s4"]), P["s4"].b, 0, 192, out, 320)
        self._maxpool(x, None, out, 512)
        return out

    def _block_e(self, x, P):
        n, h, w, _ = x.shape
        f = P["f"]
        out = torch.empty((n, h, w, 2048), dtype=x.dtype, device=x.device)
        raw = self._conv(x, f)                                       # [1x1:320 | 3x3_1:384 | dbl_1:448 | pool:192]
        self._bias_relu(raw, f.b[0:320], 0, 320, out, 0)
        t3 = self._bias_relu(raw, f.b[320:704], 320, 384)
        td = self._bias_relu(raw, f.b[704:1152], 704, 448)
        self._avgpool_bias_relu(raw, f.b[1152:1344], 1152, 192, out, 1856)
        self._bias_relu(self._conv(t3, P["a2"]), P["a2"].b, 0, 384, out, 320)
        self._bias_relu(self._conv(t3, P["b2"]), P["b2"].b, 0, 384, out, 704)
        td = self._bias_relu(self._conv(td, P["d2"]), P["d2"].b)
        self._bias_relu(self._conv(td, P["a3"]), P["a3"].b, 0, 384, out, 1088)
        self._bias_relu(self._conv(td, P["b3"]), P["b3"].b, 0, 384, out, 1472)
        return out

    @torch.no_grad()
    def __call__(self, x_nchw_channels_last):
        """Input: (N,3,299,299) tensor in channels_last memory format (what the resize kernel emits).
        Output: the feature map of the wrapper's last needed block as an NCHW *view* of NHWC storage
        (block 3: (N,2048,1,1) after the global average)."""
        x = x_nchw_channels_last.permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        a = self._bias_relu(self._conv(x, self.c1a), self.c1a.b)
        a = self._bias_relu(self._conv(a, self.c2a), self.c2a.b)
        a = self._maxpool(self._conv(a, self.c2b), self.c2b.b)                 # conv bias + ReLU + max-pool, one pass
        if self.last_block >= 1:
            a = self._bias_relu(self._conv(a, self.c3b), self.c3b.b)
            a = self._maxpool(self._conv(a, self.c4a), self.c4a.b)
        fn = {"A": self._block_a, "B": self._block_b, "C": self._block_c, "D": self._block_d, "E": self._block_e}
        for kind, P in self.blocks:
            a = fn[kind](a, P)
        if self.last_block >= 3:
            a = a.mean(dim=(1, 2), keepdim=True)                               # AdaptiveAvgPool2d((1,1))
        return a.permute(0, 3, 1, 2)


def pack_stem_mfma(w, device):
    """BatchNorm-folded stem weights (32, 3, 3, 3) [cout][cin][kh][kw] -> (fp16 (2, 32, 32) hi / lo planes in the K order of
    stem_mfma_u8_kernel, fp32 (32,) un-scaling).  Per cout a power-of-two pre-scale keeps both halves normal fp16 (as
    conv_split.SplitConv); slot u of lane half h: (h 0, u < 9) = (kh 0, t u), (0, u >= 9) = (kh 1, t u - 9), (1, u < 9) =
    (kh 2, t u), (1, u in 9, 10) = (kh 1, t 7, 8), others zero; t = 3 kw + cin; k = 16 (u // 8) + 8 h + u % 8."""
    from .conv_split import split_planes
    w = w.detach().float().cpu()
    assert tuple(w.shape) == (32, 3, 3, 3)
    amax = w.abs().reshape(32, -1).amax(1).clamp_min(1e-30)
    pre = torch.exp2(-torch.floor(torch.log2(amax)))
    wt = (w * pre.view(-1, 1, 1, 1)).permute(0, 2, 3, 1).reshape(32, 3, 9)           # [cout][kh][t = 3 kw + cin]
    wk = torch.zeros((32, 32), dtype=torch.float32)
    for h in (0, 1):
        for u in range(16):
            if h == 0:
                kh, t = (0, u) if u < 9 else (1, u - 9)
            elif u < 9:
                kh, t = 2, u
            elif u < 11:
                kh, t = 1, u - 9 + 7
            else:
                continue
            wk[:, 16 * (u // 8) + 8 * h + u % 8] = wt[:, kh, t]
    return split_planes(wk).to(device).contiguous(), (1.0 / pre).to(device).contiguous()


class SplitTrunk(FusedTrunk):
    """Same graph as ``FusedTrunk`` with every convolution after the Cin=3 stem layer on the hand-written
    split-precision fp16-MFMA kernel (``csrc/conv_split.hip``): activations travel between layers as split
    tensors (v ~= hi + lo * 2**-11, 22 mantissa bits; (N, H, W, 2C) fp16, the halves of every 32-channel block side
    by side, ``conv_split.py``), each conv's epilogue applies the folded
    BatchNorm scale/bias and ReLU and writes straight into the consumer's channel slice, the pool branch
    gets raw fp32 from the fused 1x1 conv and ``tise_avgpool3_bias_relu_split_nhwc`` finishes it.
    Measured per layer 1.5-2.3x MIOpen's fp32 kernels at a smaller error against an fp64 reference
    (``tools/conv_split_probe.py``).  The wrapper's last block decides where the forward stops (``--dims`` 64 / 192 /
    768 / 2048, inception.py:14-19): the output is always the GLOBAL MEAN of that block's feature map, (N, C, 1, 1) --
    what fid_score.py:110-111's adaptive_avg_pool2d makes of the map the reference wrapper returns."""

    def __init__(self, model, device):
        super().__init__(model, device)
        from .conv_split import SplitConv

        def sc(c):
            return SplitConv(c.w, c.b, c.stride, c.padding, self.device)

        self.s2a, self.s2b = sc(self.c2a), sc(self.c2b)
        if self.last_block >= 1:
            self.s3b, self.s4a = sc(self.c3b), sc(self.c4a)
        self.pad2b = False
        self._padbufs = {}
        # Conv2d_2a (149^2 x 32 -> 32) and Conv2d_2b (147^2 x 32 -> 64, padded): Cin = 32 means a wave can hold the whole
        # filter of 32 couts in registers (conv_pipe.hip configuration 34): the input streams through a sliding LDS ring
        # once, nothing else moves.  TISE_CONV_REGW=0: the default kernels (2a: fast, 2b: row window)
        if os.environ.get("TISE_CONV_VARIANT", "auto") in ("auto", "fast") and os.environ.get("TISE_CONV_REGW", "1") != "0":
            self.s2a = SplitConv(self.c2a.w, self.c2a.b, self.c2a.stride, self.c2a.padding, self.device, variant="pipe", pipe_cfg=34)
            # Conv2d_2b's zero padding made physical: Conv2d_2a writes into the interior of a zero-bordered buffer (out_pad)
            # and Conv2d_2b runs as a VALID convolution over it -- no per-lane tap masks in the kernel (bit-identical:
            # the masked taps contributed zeros).  TISE_CONV_PADBUF=0: the padded kernel instance on the plain tensor
            self.pad2b = os.environ.get("TISE_CONV_PADBUF", "1") != "0" and tuple(self.c2b.padding) == (1, 1)
            self.s2b = SplitConv(self.c2b.w, self.c2b.b, self.c2b.stride, (0, 0) if self.pad2b else self.c2b.padding,
                                 self.device, variant="pipe", pipe_cfg=34)
        # stem weights for the direct kernel: [kh][kw][cin][cout] fp32
        self.stem_w = self.c1a.w.permute(2, 3, 1, 0).contiguous().float()
        assert tuple(self.stem_w.shape) == (3, 3, 3, 32) and self.c1a.stride == (2, 2) and self.c1a.padding == (0, 0)
        # the stem layer on the matrix cores (csrc/conv_pipe.hip stem_mfma_u8_kernel); TISE_STEM=fma: the fp32-FMA kernel
        self.stem_mfma = os.environ.get("TISE_STEM", "mfma") == "mfma"
        self.stem_wsplit, self.stem_scale = pack_stem_mfma(self.c1a.w, self.device)
        self.sblocks = [(kind, {k: sc(v) for k, v in P.items()}) for kind, P in self.blocks]
        # the two stem max-pools are taken inside the operand load of the 1x1 convolutions that consume them
        # (Conv2d_3b; Mixed_5b's fused 1x1): conv_poolin_kernel, bit-identical to pooling first.  TISE_POOL_FUSE=0: separate kernels
        # (only with the default packing of the consuming convolutions: a TISE_CONV_VARIANT=glds / rowwin A/B run pools first)
        consumers = ([self.s3b] if self.last_block >= 1 else []) + ([self.sblocks[0][1]["f"]] if self.last_block >= 2 else [])
        self.fuse_pool = os.environ.get("TISE_POOL_FUSE", "1") != "0" and all(c.variant == "fast" for c in consumers)
        # round 4: stem max-pool 1 in the PRODUCER's epilogue -- Conv2d_2b (register-weights kernel on the zero-bordered input)
        # writes the pooled tensor only, Conv2d_3b becomes a plain 1x1 launch (conv_pipe.hip, POOL instance; bit-identical).
        # TISE_POOL_PRODUCER=0: pool 1 inside Conv2d_3b's operand load as in round 3
        self.pool_in_2b = os.environ.get("TISE_POOL_PRODUCER", "1") != "0" and os.environ.get("TISE_POOL_FUSE", "1") != "0"
        # stem max-pool 2 split between producer and consumer: Conv2d_4a (row-window kernel) takes the horizontal half in its
        # epilogue and writes 71 x 35 instead of 71 x 71 pixels, Mixed_5b's fused 1x1 takes three vertical taps instead of
        # nine in its operand load (conv_split.hip POOLH / VT; bit-identical).  TISE_POOL2_SPLIT=0: all nine taps in the consumer
        self.pool2_split = (os.environ.get("TISE_POOL2_SPLIT", "1") != "0" and self.fuse_pool and self.last_block >= 2 and
                            self.s4a.variant == "rowwin" and self.s4a.tn == 3)
        # round 5: the classifier layer (IS* logits) as a 1x1 split-precision convolution on the pool3 row -- the global-mean
        # kernel writes the row in split form as well -- with raw fp32 out (scale * conv: NO bias, which is exactly the IS*
        # head of inception_score_star_coco.py:104-105; the bird / O-IS rules add the bias afterwards).  It replaces the
        # torch / hipBLASLt GEMM, the last library kernel of the image loop.  TISE_FC=torch keeps the library GEMM.
        self.sfc = None
        self._feat_split = None
        fc = getattr(model, "fc", None)
        if (self.last_block == 3 and fc is not None and fc.in_features % 32 == 0 and os.environ.get("TISE_FC", "hip") == "hip"):
            w = fc.weight.detach().float().reshape(fc.out_features, fc.in_features, 1, 1)
            self.sfc = SplitConv(w, torch.zeros(fc.out_features), (1, 1), (0, 0), self.device, variant="fast")
            self.fc_bias = fc.bias.detach().float().to(self.device) if fc.bias is not None else None

    def fc_logits(self, n, bias=False):
        """(n, classes) fp32 logits of the pool3 rows of the LAST forward pass (their split form was kept by _global_mean)."""
        fs = self._feat_split
        assert self.sfc is not None and fs is not None and fs.shape[0] == n, "fc_logits follows a forward pass of the same batch"
        out = torch.empty((n, 1, 1, self.sfc.cout), dtype=torch.float32, device=fs.device)
        self.sfc(fs, [(0, self.sfc.cout, out, 0, 1)])
        out = out.view(n, self.sfc.cout)
        if bias and self.fc_bias is not None:
            out = out + self.fc_bias
        return out

    # ---- helpers on split tensors (N, H, W, 2C) fp16 --------------------------------------------------
    @staticmethod
    def _new(n, h, w, c, dev):
        return torch.empty((n, h, w, 2 * c), dtype=torch.float16, device=dev)

    def _sconv(self, conv, x, segs=None):
        """Run a SplitConv; with segs None the whole output goes to a fresh packed split tensor."""
        n, h, w, _ = x.shape
        oh, ow = conv.out_hw(h, w)
        if segs is None:
            out = self._new(n, oh, ow, conv.cout, x.device)
            conv(x, [(0, conv.cout, out, 0, 0)])
            return out
        conv(x, segs)
        return None

    @staticmethod
    def _maxpool_split(x, out=None, out_off=0):
        n, h, w, c2 = x.shape
        c = c2 // 2
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        if out is None:
            out = torch.empty((n, oh, ow, c2), dtype=torch.float16, device=x.device)
        _lib.call("tise_maxpool3s2_split_nhwc", _p(x), c, 0, n, h, w, c, _p(out), out.shape[3] // 2, out_off, _stream())
        return out

    @staticmethod
    def _avgpool_split(raw, bias, out, out_off):
        n, h, w, c = raw.shape
        _lib.call("tise_avgpool3_bias_relu_split_nhwc", _p(raw), c, 0, n, h, w, c, _p(bias), _p(out), out.shape[3] // 2,
                  out_off, _stream())

    def _sblock_a(self, x, P, pooled_input=False):
        n, h, w, _ = x.shape
        f = P["f"]
        if pooled_input:                                                # x is the UN-pooled tensor (stem max-pool 2 fused)
            h, w = (f.pooled_out_hw(h, w)[0], w) if pooled_input == "v" else f.pooled_out_hw(h, w)
        pf = f.cout - 176
        dev = x.device
        out = self._new(n, h, w, 224 + pf, dev)
        t5, t3 = self._new(n, h, w, 48, dev), self._new(n, h, w, 64, dev)
        raw = torch.empty((n, h, w, pf), dtype=torch.float32, device=dev)
        f(x, [(0, 64, out, 0, 0), (64, 112, t5, 0, 0), (112, 176, t3, 0, 0), (176, 176 + pf, raw, 0, 1)], pooled_input=pooled_input)
        self._avgpool_split(raw, f.bias[176:176 + pf], out, 224)
        P["c5"](t5, [(0, 64, out, 64, 0)])
        t3 = self._sconv(P["d2"], t3)
        P["d3"](t3, [(0, 96, out, 128, 0)])
        return out

    def _sblock_b(self, x, P):
        n, h, w, cin = x.shape
        cin //= 2
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        out = self._new(n, oh, ow, 384 + 96 + cin, x.device)
        P["c3"](x, [(0, 384, out, 0, 0)])
        t = self._sconv(P["d2"], self._sconv(P["d1"], x))
        P["d3"](t, [(0, 96, out, 384, 0)])
        self._maxpool_split(x, out, 480)
        return out

    def _sblock_c(self, x, P):
        n, h, w, _ = x.shape
        f = P["f"]
        c7 = (f.cout - 384) // 2
        dev = x.device
        out = self._new(n, h, w, 768, dev)
        t7, td = self._new(n, h, w, c7, dev), self._new(n, h, w, c7, dev)
        raw = torch.empty((n, h, w, 192), dtype=torch.float32, device=dev)
        f(x, [(0, 192, out, 0, 0), (192, 192 + c7, t7, 0, 0), (192 + c7, 192 + 2 * c7, td, 0, 0),
              (192 + 2 * c7, 384 + 2 * c7, raw, 0, 1)])
        self._avgpool_split(raw, f.bias[192 + 2 * c7:384 + 2 * c7], out, 576)
        t7 = self._sconv(P["s2"], t7)
        P["s3"](t7, [(0, 192, out, 192, 0)])
        for k in ("d2", "d3", "d4"):
            td = self._sconv(P[k], td)
        P["d5"](td, [(0, 192, out, 384, 0)])
        return out

    def _sblock_d(self, x, P):
        n, h, w, cin = x.shape
        cin //= 2
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        dev = x.device
        out = self._new(n, oh, ow, 320 + 192 + cin, dev)
        t3, t7 = self._new(n, h, w, 192, dev), self._new(n, h, w, 192, dev)
        P["f"](x, [(0, 192, t3, 0, 0), (192, 384, t7, 0, 0)])
        P["c32"](t3, [(0, 320, out, 0, 0)])
        for k in ("s2", "s3"):
            t7 = self._sconv(P[k], t7)
        P["s4"](t7, [(0, 192, out, 320, 0)])
        self._maxpool_split(x, out, 512)
        return out

    def _sblock_e(self, x, P):
        n, h, w, _ = x.shape
        f = P["f"]
        dev = x.device
        out = self._new(n, h, w, 2048, dev)
        t3, td = self._new(n, h, w, 384, dev), self._new(n, h, w, 448, dev)
        raw = torch.empty((n, h, w, 192), dtype=torch.float32, device=dev)
        f(x, [(0, 320, out, 0, 0), (320, 704, t3, 0, 0), (704, 1152, td, 0, 0), (1152, 1344, raw, 0, 1)])
        self._avgpool_split(raw, f.bias[1152:1344], out, 1856)
        P["a2"](t3, [(0, 384, out, 320, 0)])
        P["b2"](t3, [(0, 384, out, 704, 0)])
        td = self._sconv(P["d2"], td)
        P["a3"](td, [(0, 384, out, 1088, 0)])
        P["b3"](td, [(0, 384, out, 1472, 0)])
        return out

    @torch.no_grad()
    def __call__(self, x_nchw_channels_last):
        x = x_nchw_channels_last.permute(0, 2, 3, 1)
        if not x.is_contiguous():
            x = x.contiguous()
        n, h, w, _ = x.shape                                            # Cin = 3 stem layer: direct HIP kernel
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        a = self._new(n, oh, ow, 32, x.device)
        _lib.call("tise_stem_conv3x3s2_split", _p(x), n, h, w, _p(self.stem_w), _p(self.c1a.b), _p(a), _stream())
        return self._after_stem(a)

    @torch.no_grad()
    def forward_u8(self, u8_nhwc, lut_dev):
        """The same forward from the resized uint8 image (N,299,299,3) and the 3x256 input table on the device: the
        stem conv looks the network input values up itself (bit-identical features, 4x less input traffic)."""
        assert u8_nhwc.dtype == torch.uint8 and u8_nhwc.is_contiguous() and u8_nhwc.shape[3] == 3
        assert lut_dev.dtype == torch.float32 and lut_dev.numel() == 768 and lut_dev.is_contiguous()
        n, h, w, _ = u8_nhwc.shape
        oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
        a = self._new(n, oh, ow, 32, u8_nhwc.device)
        # the matrix-core stem reads aligned 16-byte windows: a 4-byte-aligned tensor of at least 32 bytes (a slice u8[i:j] of
        # 299 x 299 x 3 images starts on an odd address: the fp32-FMA kernel serves it, same results to the last layer's 1e-6)
        if self.stem_mfma and w >= 5 and u8_nhwc.data_ptr() % 4 == 0 and u8_nhwc.numel() >= 32:
            _lib.call("tise_stem_conv3x3s2_split_u8_mfma", _p(u8_nhwc), _p(lut_dev), n, h, w, _p(self.stem_wsplit), _p(self.stem_scale),
                      _p(self.c1a.b), _p(a), _stream())
        else:
            _lib.call("tise_stem_conv3x3s2_split_u8", _p(u8_nhwc), _p(lut_dev), n, h, w, _p(self.stem_w), _p(self.c1a.b), _p(a),
                      _stream())
        return self._after_stem(a)

    def _zero_bordered(self, n, hp, wp, c, dev):
        """Persistent split tensor (n, hp, wp, 2c) whose border stays zero: allocated and zeroed once per shape (the two
        most recent shapes are kept), only its interior is ever written.  With hipGraph replay (TISE_GRAPH=1) nothing is
        evicted: a captured graph holds the raw address of the buffer of its batch shape."""
        key = (n, hp, wp, c, str(dev))
        buf = self._padbufs.pop(key, None)
        if buf is None:
            buf = torch.zeros((n, hp, wp, 2 * c), dtype=torch.float16, device=dev)
            while len(self._padbufs) >= 2 and os.environ.get("TISE_GRAPH", "0") != "1":
                self._padbufs.pop(next(iter(self._padbufs)))
        self._padbufs[key] = buf                                        # most recently used last
        return buf

    def _after_stem(self, a):
        from .conv_split import SplitConv, pipe_fits
        n, h, w, _ = a.shape
        oh, ow = self.s2a.out_hw(h, w)
        if self.pad2b and not (pipe_fits(w, self.s2a.cout) and pipe_fits(ow + 2, self.s2b.cout)):
            # inputs wider than the sliding-window kernels' LDS ring (never the 299 x 299 network input): default kernels
            if getattr(self, "_s2b_wide", None) is None:
                self._s2b_wide = SplitConv(self.c2b.w, self.c2b.b, self.c2b.stride, self.c2b.padding, self.device, variant="fast")
            a = self._sconv(self._s2b_wide, self._sconv(self.s2a, a))
        elif self.pad2b:
            from .conv_split import pool_output_fits
            buf = self._zero_bordered(n, oh + 2, ow + 2, self.s2a.cout, a.device)
            self.s2a(a, [(0, self.s2a.cout, buf, 0, 0)], out_pad=(oh + 2, ow + 2, 1, 1))
            if self.pool_in_2b and pool_output_fits(ow + 2, ow) and oh >= 3:
                ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
                pooled = self._new(n, ph, pw, self.s2b.cout, a.device)
                self.s2b(buf, [(0, self.s2b.cout, pooled, 0, 0)], pool_output=True)     # Conv2d_2b + max-pool 1
                return self._after_pool1(pooled, True)
            a = self._sconv(self.s2b, buf)
        else:
            a = self._sconv(self.s2b, self._sconv(self.s2a, a))
        if self.last_block == 0 or not self.fuse_pool:
            return self._after_pool1(self._maxpool_split(a), True)
        return self._after_pool1(a, False)

    def _after_pool1(self, a, pooled):
        """Everything behind stem max-pool 1.  ``pooled``: ``a`` is the pooled tensor (Conv2d_2b's epilogue or the stand-alone
        kernel pooled it); otherwise ``a`` is Conv2d_2b's full result and Conv2d_3b takes the pool inside its operand load."""
        fn = {"A": self._sblock_a, "B": self._sblock_b, "C": self._sblock_c, "D": self._sblock_d, "E": self._sblock_e}
        if self.last_block == 0:                                        # --dims 64: block 0 ends with max-pool 1
            return self._global_mean(a)
        if pooled:
            t = self._sconv(self.s3b, a)                                # Conv2d_3b_1x1 on the pooled tensor
        else:
            n, h, w, _ = a.shape
            oh, ow = self.s3b.pooled_out_hw(h, w)
            t = self._new(n, oh, ow, self.s3b.cout, a.device)
            self.s3b(a, [(0, self.s3b.cout, t, 0, 0)], pooled_input=True)          # max-pool 1 + Conv2d_3b_1x1
        from .conv_split import rowwin_fits
        n, h, w, _ = t.shape
        oh, ow = self.s4a.out_hw(h, w)
        if self.pool2_split and rowwin_fits(ow, self.s4a.kw) and ow >= 3 and oh >= 3:
            hp = self._new(n, oh, (ow - 3) // 2 + 1, self.s4a.cout, t.device)
            self.s4a(t, [(0, self.s4a.cout, hp, 0, 0)], pool_h=True)               # Conv2d_4a + horizontal half of max-pool 2
            a = self._sblock_a(hp, self.sblocks[0][1], pooled_input="v")           # vertical half + Mixed_5b
            rest = self.sblocks[1:]
            for kind, P in rest:
                a = fn[kind](a, P)
            return self._global_mean(a)
        a = self._sconv(self.s4a, t)
        if self.last_block == 1:                                        # --dims 192: block 1 ends with max-pool 2
            return self._global_mean(self._maxpool_split(a))
        if self.fuse_pool:
            a = self._sblock_a(a, self.sblocks[0][1], pooled_input=True)           # max-pool 2 + Mixed_5b
            rest = self.sblocks[1:]
        else:
            a = self._maxpool_split(a)
            rest = self.sblocks
        for kind, P in rest:                                            # --dims 768 stops after Mixed_6e, 2048 after Mixed_7c
            a = fn[kind](a, P)
        return self._global_mean(a)

    def _global_mean(self, a):
        n, h, w, c2 = a.shape                                           # merge + global average, one pass
        c = c2 // 2
        feat = torch.empty((n, c), dtype=torch.float32, device=a.device)
        if self.sfc is not None and c == self.sfc.cin:                  # pool3: also as a split row, for the classifier layer
            self._feat_split = torch.empty((n, 1, 1, 2 * c), dtype=torch.float16, device=a.device)
            _lib.call("tise_split_mean_both_nhwc", _p(a), n, h * w, c, _p(feat), _p(self._feat_split), _stream())
        else:
            _lib.call("tise_split_mean_nhwc", _p(a), n, h * w, c, _p(feat), _stream())
        return feat.view(n, c, 1, 1)
