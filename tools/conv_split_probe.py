#!/usr/bin/env python3
"""Correctness + speed of the split-precision conv kernel vs torch.conv2d (MIOpen fp32) on trunk shapes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
from tise_toolbox_amd.conv_split import SplitConv, merge, split  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
shapes = [  # (H, W, Cin, Cout, kh, kw, stride, pad)
    (73, 73, 80, 192, 3, 3, 1, (0, 0)), (147, 147, 32, 64, 3, 3, 1, (1, 1)), (35, 35, 288, 384, 3, 3, 2, (0, 0)),
    (8, 8, 2048, 1344, 1, 1, 1, (0, 0)), (17, 17, 768, 704, 1, 1, 1, (0, 0)), (35, 35, 96, 96, 3, 3, 1, (1, 1)),
    (17, 17, 160, 160, 1, 7, 1, (0, 3)), (17, 17, 192, 192, 7, 1, 1, (3, 0)), (35, 35, 48, 64, 5, 5, 1, (2, 2)),
    (8, 8, 448, 384, 3, 3, 1, (1, 1)), (35, 35, 192, 208, 1, 1, 1, (0, 0)), (8, 8, 384, 384, 1, 3, 1, (0, 1)),
]
g = torch.Generator(device="cpu").manual_seed(0)
tot_m = tot_s = tot_g = tot_g3 = 0.0
for (H, W, Cin, Cout, kh, kw, st, pad) in shapes:
    x = torch.rand((B, H, W, Cin), generator=g).to(dev) * 2.0                       # non-negative like post-ReLU
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    conv = SplitConv(w, b, (st, st), pad, dev)
    conv.variant = "reg"
    xs = split(x)
    oh, ow = conv.out_hw(H, W)
    out = torch.empty((2, B, oh, ow, Cout), dtype=torch.float16, device=dev)
    conv(xs, [(0, Cout, out, 0, 0)])
    wcl = w.contiguous(memory_format=torch.channels_last)
    ref = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2), wcl, b, st, pad)).permute(0, 2, 3, 1)
    # fp64 reference on a subset for an unbiased error measure
    nref = min(B, 4)
    ref64 = torch.relu(torch.conv2d(x[:nref].permute(0, 3, 1, 2).double(), w.double(), b.double(), st, pad)).permute(0, 2, 3, 1)
    got = merge(out)
    e_split = (got[:nref].double() - ref64).abs().max().item()
    e_mi = (ref[:nref].double() - ref64).abs().max().item()
    scale = ref64.abs().max().item()

    def t(fn, it=5):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it
    ms_s = t(lambda: conv(xs, [(0, Cout, out, 0, 0)]))
    res = {}
    for var in ("glds", "fast"):
        conv.variant = var
        out2 = torch.zeros_like(out)
        same = True
        for rep in range(4):                       # repeated runs: a pipeline race shows up as a mismatch
            out2.zero_()
            conv(xs, [(0, Cout, out2, 0, 0)])
            same = same and bool(torch.equal(out2, out))
        res[var] = (t(lambda: conv(xs, [(0, Cout, out2, 0, 0)])), same)
    conv.variant = "reg"
    ms_m = t(lambda: torch.conv2d(x.permute(0, 3, 1, 2), wcl, None, st, pad))
    flop = 2.0 * B * oh * ow * Cout * Cin * kh * kw
    tot_m += ms_m; tot_s += ms_s; tot_g += res['glds'][0]; tot_g3 += res['fast'][0]
    print(f"{H}x{W}x{Cin}->{Cout} k{kh}x{kw} s{st} tn{conv.tn}: split {ms_s:7.3f} ms ({flop/ms_s/1e9:6.1f} TF-eq, {3*flop/ms_s/1e9:6.0f} TF fp16)  "
          f"miopen {ms_m:7.3f} ms ({flop/ms_m/1e9:6.1f} TF)  speedup {ms_m/ms_s:4.2f}x  glds {res['glds'][0]:6.3f} ms ({3*flop/res['glds'][0]/1e9:4.0f} TF16 same={res['glds'][1]})  fast {res['fast'][0]:6.3f} ms ({3*flop/res['fast'][0]/1e9:4.0f} TF16 same={res['fast'][1]})  err split {e_split/scale:.2e} miopen {e_mi/scale:.2e}", flush=True)
print(f"total: reg {tot_s:.2f} ms, glds {tot_g:.2f} ms, fast {tot_g3:.2f} ms, miopen {tot_m:.2f} ms")
