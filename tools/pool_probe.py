#!/usr/bin/env python3
"""3x3 / stride-2 max pool on split planes at the trunk's four shapes (batch 500), launched back to back on the same
input: time per launch, effective HBM rate (one read of the input + one write of the output) and bit-exactness
against torch on the merged fp32 values.  Result (profiles/r02b_pool_probe.txt): 4.7-4.9 TB/s at every shape, also at
147 x 147 x 64 (0.73 ms, the same as the steady-state launch inside the trunk; the 1.34 ms / 2.57 TB/s quoted in round 1
was the MAXIMUM over calls of the rocprof table, a cold first launch).
A 2x2-outputs-per-thread variant (25 instead of 36 loads per plane) was slower (3.4 TB/s: the lexicographic fp16
compares cost more than the loads they save) and was dropped."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import merge, split  # noqa: E402
from tise_toolbox_amd.trunk import SplitTrunk  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for h, c in ((147, 64), (71, 192), (35, 288), (17, 768)):
    x32 = torch.relu(torch.randn((B, h, h, c), device=dev)) * 3.0
    x = split(x32).contiguous()
    out = SplitTrunk._maxpool_split(x)
    want = torch.nn.functional.max_pool2d(merge(x).permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    ok = torch.equal(merge(out), want)
    for _ in range(3):
        SplitTrunk._maxpool_split(x, out)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        SplitTrunk._maxpool_split(x, out)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    gb = (x.numel() + out.numel()) * 2 / 1e9
    print(f"{h}x{h}x{c}: {ms:.3f} ms  {gb / ms:.2f} TB/s  exact={ok}")
    del x, x32, out, want
