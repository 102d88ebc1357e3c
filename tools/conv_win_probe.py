#!/usr/bin/env python3
"""Window-resident conv kernel vs the per-tap DMA kernel on the trunk's stride-1 multi-tap layers:
accuracy against an fp64 convolution (small batch) and time per launch (full batch)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split, merge

LAYERS = [("2a", 149, 149, 32, 32, 3, 3, (0, 0)), ("2b", 147, 147, 32, 64, 3, 3, (1, 1)),
          ("4a", 73, 73, 80, 192, 3, 3, (0, 0)), ("5x5", 35, 35, 48, 64, 5, 5, (2, 2)),
          ("3x3a", 35, 35, 64, 96, 3, 3, (1, 1)), ("3x3b", 35, 35, 96, 96, 3, 3, (1, 1)),
          ("1x7_128", 17, 17, 128, 128, 1, 7, (0, 3)), ("7x1_160", 17, 17, 160, 192, 7, 1, (3, 0)),
          ("1x7_192", 17, 17, 192, 192, 1, 7, (0, 3)), ("1x3", 8, 8, 384, 384, 1, 3, (0, 1)),
          ("3x1", 8, 8, 384, 384, 3, 1, (1, 0)), ("3x3_448", 8, 8, 448, 384, 3, 3, (1, 1))]
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
for name, H, W, Cin, Cout, kh, kw, pad in LAYERS:
    if only and name not in only:
        continue
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    res = {}
    for var in ("fast", "win"):
        for tn in ([None] if var == "fast" else sorted({1, 2, 3, 4, 5} & {t for t in range(1, 6) if 32 * t <= max(32, -(-Cout // 32) * 32)})):
            conv = SplitConv(w, b, (1, 1), pad, dev, tn=tn, variant=var)
            oh, ow = conv.out_hw(H, W)
            # accuracy, small batch
            xs_small = (torch.rand((3, H, W, Cin), generator=g) * 3.0).to(dev)
            out = torch.zeros((2, 3, oh, ow, Cout), dtype=torch.float16, device=dev)
            try:
                conv(split(xs_small), [(0, Cout, out, 0, 0)])
            except Exception as e:
                print(name, var, tn, "ERR", e); continue
            ref = torch.relu(torch.conv2d(xs_small.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, pad)).permute(0, 2, 3, 1)
            err = (merge(out).double() - ref).abs().max().item() / ref.abs().max().item()
            x = split((torch.rand((N, H, W, Cin), device=dev) * 3.0))
            out = torch.zeros((2, N, oh, ow, Cout), dtype=torch.float16, device=dev)
            for _ in range(3):
                conv(x, [(0, Cout, out, 0, 0)])
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                conv(x, [(0, Cout, out, 0, 0)])
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / 10
            tf = 2.0 * N * oh * ow * Cout * Cin * kh * kw / ms / 1e9
            print(f"{name:8s} {var:4s} tn={conv.tn} err={err:.2e} {ms:7.3f} ms {tf:6.1f} TF", flush=True)
            del x, out
