#!/usr/bin/env python3
"""One conv layer, a few launches per variant, for rocprofv3 --pmc passes."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import SplitConv, split  # noqa: E402

dev = torch.device("cuda", 0)
B = 500
H, W, Cin, Cout, kh, kw, st, pad = [int(v) for v in (sys.argv[1:9] if len(sys.argv) > 8 else (35, 35, 288, 384, 3, 3, 2, 0))]
g = torch.Generator(device="cpu").manual_seed(0)
x = (torch.rand((B, H, W, Cin), generator=g) * 2).to(dev)
w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
b = torch.zeros(Cout, device=dev)
conv = SplitConv(w, b, (st, st), (pad, pad), dev)
xs = split(x)
oh, ow = conv.out_hw(H, W)
out = torch.empty((2, B, oh, ow, Cout), dtype=torch.float16, device=dev)
for var in ("reg", "glds", "gldsb", "glds3"):
    conv.variant = var
    for _ in range(3):
        conv(xs, [(0, Cout, out, 0, 0)])
torch.cuda.synchronize()
print("done")
