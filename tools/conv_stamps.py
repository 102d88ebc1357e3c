#!/usr/bin/env python3
"""In-kernel stamps of the persistent conv kernel: where one K-step's cycles go (waves 0 and 4 of workgroup 0)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, ctypes
import numpy as np
from tise_toolbox_amd.conv_split import SplitConv, split, ConvArgs
from tise_toolbox_amd import _lib

dev = torch.device("cuda:0")
H, Cin, Cout, kh, kw, st, pad = 35, 288, 384, 3, 3, 2, (0, 0)
if len(sys.argv) > 2:
    H, Cin, Cout, kh, kw, st = [int(v) for v in sys.argv[2].split(",")]
    pad = (kh // 2, kw // 2) if st == 1 else (0, 0)
cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
flags = int(sys.argv[3], 0) if len(sys.argv) > 3 else 0
N = 500
g = torch.Generator(device="cpu").manual_seed(1)
w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
conv = SplitConv(w, b, (st, st), pad, dev, variant="pipe", pipe_cfg=cfg)
oh, ow = conv.out_hw(H, H)
x = split((torch.rand((N, H, H, Cin), device=dev) * 3.0))
out = torch.zeros((2, N, oh, ow, Cout), dtype=torch.float16, device=dev)
stamp = torch.zeros(1024, dtype=torch.int32, device=dev)
for _ in range(3):
    conv(x, [(0, Cout, out, 0, 0)])
conv.debug_flags = 0x800 | flags
conv.debug_ptr = stamp.data_ptr()
conv(x, [(0, Cout, out, 0, 0)])
torch.cuda.synchronize()
s = stamp.cpu().numpy().astype(np.int64) & 0xffffffff
for wv in (0, 1):
    t = s[wv * 512: wv * 512 + 480].reshape(96, 5)
    print(f"wave {wv * 4}: step  wait  barrier  issue  compute  | step total   (cycles of s_memtime)")
    for i in range(2, 40):
        d = [(t[i, k + 1] - t[i, k]) & 0xffffffff for k in range(4)]
        tot = (t[i + 1, 0] - t[i, 0]) & 0xffffffff
        print(f"   {i:3d}  {d[0]:6d} {d[1]:6d} {d[2]:6d} {d[3]:6d}   | {tot:6d}")
    tt = (t[60, 0] - t[20, 0]) & 0xffffffff
    rt = (int(s[wv * 512 + 481]) - int(s[wv * 512 + 480])) & 0xffffffff
    print(f"   mean step (20..60): {tt / 40:.0f} ticks; wall {rt * 10} ns for 40 steps -> {tt / (rt * 10.0):.3f} ticks/ns, {rt * 10 / 40:.0f} ns/step")
