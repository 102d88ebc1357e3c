#!/usr/bin/env python3
"""Does the default conv kernel's speed depend on the DATA (i.e. on power / clock), not only on the instruction stream?
Same layer, same launch, inputs: all zeros | ReLU(randn) (half zeros) | dense random.  Batch 500, median of 5 x 5 launches."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import SplitConv, split  # noqa: E402

dev = torch.device("cuda", 0)
B = 500
for H, Cin, Cout, kh, kw, st, pad in [(17, 768, 768, 1, 1, 1, (0, 0)), (17, 192, 192, 1, 7, 1, (0, 3)), (35, 96, 96, 3, 3, 1, (1, 1)),
                                      (8, 2048, 1344, 1, 1, 1, (0, 0))]:
    g = torch.Generator(device="cpu").manual_seed(0)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = torch.zeros(Cout, device=dev)
    conv = SplitConv(w, b, (st, st), pad, dev)
    oh, ow = conv.out_hw(H, H)
    out = torch.zeros((B, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
    flop = 3 * 2.0 * B * oh * ow * Cout * Cin * kh * kw
    line = f"{H}x{H}x{Cin}->{Cout} k{kh}x{kw}:"
    base = torch.randn((B, H, H, Cin), generator=g).to(dev)
    for name, x in (("zeros", torch.zeros_like(base)), ("relu(randn)", torch.relu(base)), ("dense", base.abs() + 0.5)):
        xs = split(x)
        for _ in range(3):
            conv(xs, [(0, Cout, out, 0, 0)])
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                conv(xs, [(0, Cout, out, 0, 0)])
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 5)
        t = sorted(ts)[2]
        line += f"  {name} {t:.3f} ms {flop / t / 1e9:5.0f} TF16"
    print(line, flush=True)
