#!/usr/bin/env python3
"""Stand-alone launch times of the trunk's non-convolution kernels at batch 1000 (stem, pools, global mean): A/B of two
builds of the library through TISE_LIB_PATH."""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tise_toolbox_amd import _lib, device  # noqa: E402
from tise_toolbox_amd.trunk import SplitTrunk  # noqa: E402

dev = torch.device("cuda", 0)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def timed(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def st():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def p(t):
    return ctypes.c_void_p(t.data_ptr())


u8 = torch.randint(0, 256, (N, 299, 299, 3), dtype=torch.uint8, device=dev)
lut = torch.rand(768, device=dev)
w = torch.randn(27 * 32, device=dev)
b = torch.randn(32, device=dev)
out = torch.empty((N, 149, 149, 64), dtype=torch.float16, device=dev)
print(f"stem u8            {timed(lambda: _lib.call('tise_stem_conv3x3s2_split_u8', p(u8), p(lut), N, 299, 299, p(w), p(b), p(out), st())):7.3f} ms")
for (h, wd, c, co) in ((35, 35, 32, 256), (35, 35, 64, 288), (17, 17, 192, 768), (8, 8, 192, 2048)):
    raw = torch.rand((N, h, wd, c), device=dev)
    bias = torch.rand(c, device=dev)
    o = torch.empty((N, h, wd, 2 * co), dtype=torch.float16, device=dev)
    print(f"avgpool {h}x{wd}x{c:<4d}  {timed(lambda: SplitTrunk._avgpool_split(raw, bias, o, co - c)):7.3f} ms")
for (h, wd, c, co) in ((35, 35, 288, 768), (17, 17, 768, 1280), (147, 147, 64, 64)):
    n = N if h < 100 else N // 4
    x = torch.rand((n, h, wd, 2 * c), device=dev).half()
    oh, ow = (h - 3) // 2 + 1, (wd - 3) // 2 + 1
    o = torch.empty((n, oh, ow, 2 * co), dtype=torch.float16, device=dev)
    print(f"maxpool {h}x{wd}x{c:<4d} n={n} {timed(lambda: SplitTrunk._maxpool_split(x, o, co - c)):7.3f} ms")
a = torch.rand((N, 8, 8, 4096), device=dev).half()
f = torch.empty((N, 2048), device=dev)
print(f"split_mean         {timed(lambda: _lib.call('tise_split_mean_nhwc', p(a), N, 64, 2048, p(f), st())):7.3f} ms")
print(f"resize u8          {timed(lambda: device.resize_u8_only(u8[:, :256, :256].contiguous(), (299, 299))):7.3f} ms")
from tise_toolbox_amd.trunk import pack_stem_mfma  # noqa: E402
wsp, sc = pack_stem_mfma(w.view(3, 3, 3, 32).permute(3, 2, 0, 1).contiguous(), dev)
print(f"stem u8 (MFMA)     {timed(lambda: _lib.call('tise_stem_conv3x3s2_split_u8_mfma', p(u8), p(lut), N, 299, 299, p(wsp), p(sc), p(b), p(out), st())):7.3f} ms")
