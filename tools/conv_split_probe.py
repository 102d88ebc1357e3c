#!/usr/bin/env python3
"""Correctness + speed of the split-precision conv kernels (default `fast`, generic `glds`) vs torch.conv2d (MIOpen
fp32) on trunk shapes: time per launch, error of each against an fp64 convolution, repeatability."""
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


def timed(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


g = torch.Generator(device="cpu").manual_seed(0)
tot = {"fast": 0.0, "glds": 0.0, "miopen": 0.0}
for (H, W, Cin, Cout, kh, kw, st, pad) in shapes:
    x = torch.rand((B, H, W, Cin), generator=g).to(dev) * 2.0                       # non-negative like post-ReLU
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.1).to(dev)
    xs = split(x)
    wcl = w.contiguous(memory_format=torch.channels_last)
    ref = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2), wcl, b, st, pad)).permute(0, 2, 3, 1)
    nref = min(B, 4)                                                                # fp64 reference on a subset
    ref64 = torch.relu(torch.conv2d(x[:nref].permute(0, 3, 1, 2).double(), w.double(), b.double(), st, pad)).permute(0, 2, 3, 1)
    scale = ref64.abs().max().item()
    e_mi = (ref[:nref].double() - ref64).abs().max().item() / scale
    res = {}
    for var in ("fast", "glds"):
        conv = SplitConv(w, b, (st, st), pad, dev, variant=var)
        oh, ow = conv.out_hw(H, W)
        out = torch.zeros((B, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
        conv(xs, [(0, Cout, out, 0, 0)])
        first = out.clone()
        same = True
        for rep in range(3):                       # repeated runs: a pipeline race shows up as a mismatch
            out.zero_()
            conv(xs, [(0, Cout, out, 0, 0)])
            same = same and bool(torch.equal(out, first))
        err = (merge(out)[:nref].double() - ref64).abs().max().item() / scale
        res[var] = (timed(lambda: conv(xs, [(0, Cout, out, 0, 0)])), err, same, conv.tn)
        tot[var] += res[var][0]
    ms_m = timed(lambda: torch.conv2d(x.permute(0, 3, 1, 2), wcl, None, st, pad))
    tot["miopen"] += ms_m
    flop = 2.0 * B * oh * ow * Cout * Cin * kh * kw
    f, gl = res["fast"], res["glds"]
    print(f"{H}x{W}x{Cin}->{Cout} k{kh}x{kw} s{st} tn{f[3]}: fast {f[0]:7.3f} ms ({flop/f[0]/1e9:6.1f} TF-eq, {3*flop/f[0]/1e9:6.0f} TF fp16, "
          f"err {f[1]:.2e}, repeatable={f[2]})  glds {gl[0]:7.3f} ms (err {gl[1]:.2e}, repeatable={gl[2]})  "
          f"miopen fp32 {ms_m:7.3f} ms ({flop/ms_m/1e9:6.1f} TF, err {e_mi:.2e})  speedup {ms_m/f[0]:4.2f}x", flush=True)
print(f"total: fast {tot['fast']:.2f} ms, glds {tot['glds']:.2f} ms, miopen {tot['miopen']:.2f} ms")
