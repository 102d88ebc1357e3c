#!/usr/bin/env python3
"""In-kernel stamps of the default conv kernel (conv_split_fast_kernel): where one K-step's cycles go, waves 0-3 of
workgroup 0.   usage: conv_stamps.py [H,Cin,Cout,kh,kw,stride]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
import numpy as np
from tise_toolbox_amd.conv_split import SplitConv, split, ConvArgs
from tise_toolbox_amd import _lib

dev = torch.device("cuda:0")
H, Cin, Cout, kh, kw, st, pad = 35, 288, 384, 3, 3, 2, (0, 0)
if len(sys.argv) > 1:
    H, Cin, Cout, kh, kw, st = [int(v) for v in sys.argv[1].split(",")]
    pad = (kh // 2, kw // 2) if st == 1 else (0, 0)
flags = int(sys.argv[2], 0) if len(sys.argv) > 2 else 0
N = 500
g = torch.Generator(device="cpu").manual_seed(1)
w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
conv = SplitConv(w, b, (st, st), pad, dev)
oh, ow = conv.out_hw(H, H)
x = split((torch.rand((N, H, H, Cin), device=dev) * 3.0))
out = torch.zeros((N, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
stamp = torch.zeros(2048, dtype=torch.int32, device=dev)
for _ in range(3):
    conv(x, [(0, Cout, out, 0, 0)])
def launch_ms(f, it=10):
    conv.debug_flags = f
    conv(x, [(0, Cout, out, 0, 0)])
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        conv(x, [(0, Cout, out, 0, 0)])
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


conv.debug_ptr = stamp.data_ptr()
print(f"launch: product kernel {launch_ms(0):.3f} ms, instrumented {launch_ms(0x800):.3f} ms, "
      f"instrumented with the pixel operand fetched every 7th step only (0x1000) {launch_ms(0x1800):.3f} ms, "
      f"no DMA at all (0x100) {launch_ms(0x900):.3f} ms, no MFMAs (0x200) {launch_ms(0xa00):.3f} ms")
conv.debug_flags = 0x800 | flags
conv.debug_ptr = stamp.data_ptr()
conv(x, [(0, Cout, out, 0, 0)])
torch.cuda.synchronize()
s = stamp.cpu().numpy().astype(np.int64) & 0xffffffff
nsteps = min(94, conv.kpad // 32)
for wv in range(4):
    t = s[wv * 512: wv * 512 + 480].reshape(96, 5)
    d = np.array([[(t[i, k + 1] - t[i, k]) & 0xffffffff for k in range(4)] + [(t[i + 1, 0] - t[i, 0]) & 0xffffffff]
                  for i in range(2, nsteps - 1)], dtype=np.float64)
    rt = (int(s[wv * 512 + 481]) - int(s[wv * 512 + 480])) & 0xffffffff          # s_memrealtime: 100 MHz
    span = (t[nsteps - 1, 4] - t[0, 0]) & 0xffffffff
    print(f"wave {wv}: mean over steps 2..{nsteps - 2}:  wait-DMA {d[:, 0].mean():6.0f}  barrier {d[:, 1].mean():6.0f}  issue-DMA {d[:, 2].mean():6.0f}  "
          f"issue-reads+MFMAs {d[:, 3].mean():6.0f}  | step {d[:, 4].mean():6.0f} ticks;  K loop {span} ticks in {rt * 10} ns "
          f"-> {span / max(rt * 10.0, 1):.3f} ticks/ns")
    if wv == 0:
        for i in range(0, min(12, len(d))):
            print("      step", i + 2, " ".join(f"{int(v):6d}" for v in d[i]))
