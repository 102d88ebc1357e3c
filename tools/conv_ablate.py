#!/usr/bin/env python3
"""Ablation of the default conv kernel on trunk layers: time with the DMA, the MFMA phase or the epilogue
switched off (results are then garbage; only the time matters)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split

LAYERS = [("2bv", 149, 32, 64, 3, 3, 1, (0, 0))] if (len(sys.argv) > 2 and sys.argv[2] == "34") else [("2b", 147, 32, 64, 3, 3, 1, (1, 1)), ("5c3x3", 35, 96, 96, 3, 3, 1, (1, 1)),
          ("6b1x1", 17, 768, 704, 1, 1, 1, (0, 0)), ("7c3x3", 8, 448, 384, 3, 3, 1, (1, 1)),
          ("5b1x1", 35, 192, 208, 1, 1, 1, (0, 0)), ("6a", 35, 288, 384, 3, 3, 2, (0, 0)),
          ("6e7x1", 17, 192, 192, 7, 1, 1, (3, 0))]
dev = torch.device("cuda:0")
VARIANT = sys.argv[1] if len(sys.argv) > 1 else "fast"
CFG = int(sys.argv[2]) if len(sys.argv) > 2 else None
N = 500
for name, H, Cin, Cout, kh, kw, st, pad in LAYERS:
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (st, st), pad, dev, variant=VARIANT, pipe_cfg=CFG)
    oh, ow = conv.out_hw(H, H)
    x = split((torch.rand((N, H, H, Cin), device=dev) * 3.0))
    out = torch.zeros((N, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
    line = f"{name:7s} {VARIANT} tn={conv.tn} cfg={conv.pipe_cfg} K={conv.k:5d}"
    for label, flags in (("full", 0), ("noDMA", 0x100), ("noMMA", 0x200), ("noEPI", 0x400), ("noDMA+noEPI", 0x500),
                         ("noMMA+noEPI", 0x600), ("onlyEPI", 0x300), ("nothing", 0x700), ("noSTORE", 0x2000), ("onlyEPInoSTORE", 0x2300)):
        conv.debug_flags = flags
        for _ in range(3):
            conv(x, [(0, Cout, out, 0, 0)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(x, [(0, Cout, out, 0, 0)])
        e1.record(); torch.cuda.synchronize()
        line += f"  {label} {e0.elapsed_time(e1) / 10:6.3f}"
    print(line, flush=True)
