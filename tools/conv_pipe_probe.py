#!/usr/bin/env python3
"""Persistent 3-stage conv kernel (conv_pipe.hip) vs the default 2-stage kernel on every distinct trunk
conv shape: accuracy against fp64 (small batch) and time per launch (batch 500)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split, merge, PIPE_BN

# (H, Cin, Cout, kh, kw, stride, pad, launches per trunk forward)
LAYERS = [(73, 80, 192, 3, 3, 1, (0, 0), 1), (147, 32, 64, 3, 3, 1, (1, 1), 1), (35, 96, 96, 3, 3, 1, (1, 1), 3),
          (17, 768, 704, 1, 1, 1, (0, 0), 2), (149, 32, 32, 3, 3, 1, (0, 0), 1), (35, 64, 96, 3, 3, 1, (1, 1), 4),
          (35, 48, 64, 5, 5, 1, (2, 2), 3), (17, 192, 192, 1, 7, 1, (0, 3), 4), (17, 192, 192, 7, 1, 1, (3, 0), 4),
          (35, 288, 384, 3, 3, 2, (0, 0), 1), (17, 160, 160, 1, 7, 1, (0, 3), 4), (17, 160, 160, 7, 1, 1, (3, 0), 4),
          (17, 768, 768, 1, 1, 1, (0, 0), 1), (8, 448, 384, 3, 3, 1, (1, 1), 2), (35, 288, 240, 1, 1, 1, (0, 0), 1),
          (8, 2048, 1344, 1, 1, 1, (0, 0), 1), (73, 64, 80, 1, 1, 1, (0, 0), 1), (17, 768, 640, 1, 1, 1, (0, 0), 1),
          (35, 256, 240, 1, 1, 1, (0, 0), 1), (17, 160, 192, 7, 1, 1, (3, 0), 2), (17, 160, 192, 1, 7, 1, (0, 3), 2),
          (35, 192, 208, 1, 1, 1, (0, 0), 1), (8, 384, 384, 1, 3, 1, (0, 1), 4), (8, 384, 384, 3, 1, 1, (1, 0), 4),
          (8, 1280, 1344, 1, 1, 1, (0, 0), 1), (17, 768, 384, 1, 1, 1, (0, 0), 1), (17, 128, 128, 1, 7, 1, (0, 3), 2),
          (17, 128, 128, 7, 1, 1, (3, 0), 2), (35, 288, 64, 1, 1, 1, (0, 0), 1), (17, 128, 192, 7, 1, 1, (3, 0), 1),
          (17, 128, 192, 1, 7, 1, (0, 3), 1), (35, 96, 96, 3, 3, 2, (0, 0), 1), (17, 192, 320, 3, 3, 2, (0, 0), 1),
          (17, 192, 192, 3, 3, 2, (0, 0), 1)]
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
cfgs = [int(c) for c in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0, 1, 2, 3, 6]
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
tot_fast = tot_best = 0.0
for li, (H, Cin, Cout, kh, kw, st, pad, cnt) in enumerate(LAYERS):
    if only is not None and li != only:
        continue
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    xs_small = (torch.rand((3, H, H, Cin), generator=g) * 3.0).to(dev)
    ref = torch.relu(torch.conv2d(xs_small.permute(0, 3, 1, 2).double(), w.double(), b.double(), st, pad)).permute(0, 2, 3, 1)
    x = split((torch.rand((N, H, H, Cin), device=dev) * 3.0))
    res = []
    for var, cfg in [("fast", None)] + [("pipe", c) for c in cfgs]:
        if cfg is not None and PIPE_BN[cfg] >= 2 * max(32, Cout):
            continue
        if cfg is not None and ((cfg >= 11 and cfg < 33) or cfg == 7) and (st != 1 or kh * kw == 1):
            continue
        if cfg is not None and cfg >= 40 and Cin % 32 != 0:
            continue
        conv = SplitConv(w, b, (st, st), pad, dev, variant=var, pipe_cfg=cfg)
        oh, ow = conv.out_hw(H, H)
        out = torch.zeros((2, 3, oh, ow, Cout), dtype=torch.float16, device=dev)
        conv(split(xs_small), [(0, Cout, out, 0, 0)])
        err = (merge(out).double() - ref).abs().max().item() / ref.abs().max().item()
        out = torch.zeros((2, N, oh, ow, Cout), dtype=torch.float16, device=dev)
        for _ in range(3):
            conv(x, [(0, Cout, out, 0, 0)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(x, [(0, Cout, out, 0, 0)])
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        tf = 3 * 2.0 * N * oh * ow * Cout * Cin * kh * kw / ms / 1e9
        res.append((var if cfg is None else f"p{cfg}", ms, tf, err))
        del out
    fast = res[0][1]
    best = min(r[1] for r in res)
    tot_fast += fast * cnt; tot_best += best * cnt
    print(f"{H}x{H}x{Cin}->{Cout} k{kh}x{kw} s{st} x{cnt}: " +
          "  ".join(f"{n} {ms:.3f}ms {tf:4.0f}TF e={err:.1e}" for n, ms, tf, err in res), flush=True)
    del x
print(f"trunk total: fast {tot_fast:.2f} ms, best-of {tot_best:.2f} ms")
