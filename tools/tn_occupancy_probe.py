#!/usr/bin/env python3
"""Does a third resident workgroup per CU pay?  The default conv kernel on deep 1x1 / 7x1 trunk layers at TN = 2 (128 x 64 tiles, 155
VGPRs, 48 KB of LDS: THREE workgroups per CU) against the product's TN (128 x 128 / 128 x 96 / 128 x 160: two per CU), batch 3000."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split, pick_tn

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
for name, H, Cin, Cout, kh, kw in (("17x17x768->704 1x1", 17, 768, 704, 1, 1), ("17x17x768->768 1x1", 17, 768, 768, 1, 1), ("8x8x2048->1344 1x1", 8, 2048, 1344, 1, 1),
                                   ("17x17x160->160 7x1", 17, 160, 160, 7, 1), ("17x17x192->192 7x1", 17, 192, 192, 7, 1), ("35x35x288->240 1x1", 35, 288, 240, 1, 1)):
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    x = split(torch.relu(torch.randn((N, H, H, Cin), device=dev)))
    out = torch.zeros((N, H, H, 2 * Cout), dtype=torch.float16, device=dev)
    line = f"{name:22s}"
    for tn in (pick_tn(Cout), 2, 3, 4):
        conv = SplitConv(w, b, (1, 1), (kh // 2, kw // 2), dev, tn=tn, variant="fast")
        for _ in range(2):
            conv(x, [(0, Cout, out, 0, 0)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            conv(x, [(0, Cout, out, 0, 0)])
        e1.record(); torch.cuda.synchronize()
        line += f"   tn{tn}: {e0.elapsed_time(e1) / 5:7.3f} ms"
    print(line, flush=True)
