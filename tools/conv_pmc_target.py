#!/usr/bin/env python3
"""A few launches of chosen conv kernels on trunk layers, for rocprofv3 --pmc passes.
usage: conv_pmc_target.py  (no arguments)  -- kernels are identified by name + grid size in the counter CSV."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import SplitConv, split  # noqa: E402

dev = torch.device("cuda", 0)
B = 500
LAYERS = [("5c3x3", 35, 96, 96, 3, 3, 1, (1, 1)), ("6b1x1", 17, 768, 704, 1, 1, 1, (0, 0)), ("6a", 35, 288, 384, 3, 3, 2, (0, 0))]
for name, H, Cin, Cout, kh, kw, st, pad in LAYERS:
    g = torch.Generator(device="cpu").manual_seed(0)
    x = (torch.rand((B, H, H, Cin), generator=g) * 2).to(dev)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = torch.zeros(Cout, device=dev)
    xs = split(x)
    for var, cfg in (("fast", None), ("pipe", 0), ("pipe", 8)):
        conv = SplitConv(w, b, (st, st), pad, dev, variant=var, pipe_cfg=cfg)
        oh, ow = conv.out_hw(H, H)
        out = torch.empty((B, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
        for _ in range(3):
            conv(xs, [(0, Cout, out, 0, 0)])
        torch.cuda.synchronize()
        print(name, var, cfg, "launched")
print("done")
