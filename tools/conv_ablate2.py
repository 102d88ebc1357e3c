#!/usr/bin/env python3
"""Where the epilogue's in-kernel cost comes from: the default conv kernel on two trunk layers with the K loop's halves and the
epilogue's halves switched off in combination (switches 0x100 no DMA, 0x200 no MFMA, 0x400 no epilogue, 0x2000 epilogue without
its global stores).  ms per 500 images; results are garbage, only the time matters."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split

dev = torch.device("cuda:0")
N = 500
for name, H, Cin, Cout in (("17x17x768->704", 17, 768, 704), ("35x35x192->208", 35, 192, 208), ("8x8x2048->1344", 8, 2048, 1344)):
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (1, 1), (0, 0), dev, variant="fast")
    x = split((torch.rand((N, H, H, Cin), device=dev) * 3.0))
    out = torch.zeros((N, H, H, 2 * Cout), dtype=torch.float16, device=dev)
    line = f"{name:16s} tn={conv.tn}"
    for label, flags in (("full", 0), ("K loop only", 0x400), ("K loop + conversion (no stores)", 0x2000),
                         ("MFMA + epilogue", 0x100), ("MFMA + conversion", 0x2100), ("MFMA only", 0x500),
                         ("DMA + epilogue", 0x200), ("DMA + conversion", 0x2200), ("DMA only", 0x600),
                         ("epilogue only", 0x300), ("conversion only", 0x2300), ("nothing", 0x700)):
        conv.debug_flags = flags
        for _ in range(3):
            conv(x, [(0, Cout, out, 0, 0)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            conv(x, [(0, Cout, out, 0, 0)])
        e1.record(); torch.cuda.synchronize()
        line += f"\n    {label:34s} {e0.elapsed_time(e1) / 20:6.3f}"
    print(line, flush=True)
