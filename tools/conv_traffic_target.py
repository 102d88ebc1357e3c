#!/usr/bin/env python3
"""Three forward passes of the split-precision trunk at batch 500 for rocprofv3 --pmc passes (FETCH_SIZE / WRITE_SIZE,
and the SQ_* issue counters); prints -- and writes to gpurun_out/conv_traffic_layers.json -- the algorithmic bytes of
the conv launches of one pass (input + output activations + weights).  Condensed by tools/conv_pmc_summary.py."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import SplitConv  # noqa: E402
from tise_toolbox_amd.inception import InceptionV3  # noqa: E402
from tise_toolbox_amd.trunk import SplitTrunk  # noqa: E402

dev = torch.device("cuda", 0)
B = 500
trunk = SplitTrunk(InceptionV3([3], seed=0), dev)
x = torch.rand((B, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
orig = SplitConv.__call__
alg = []
detail = []


def wrapped(self, xs, segs, pooled_input=False, **kw):
    r = orig(self, xs, segs, pooled_input=pooled_input, **kw)
    n, h, w, _ = xs.shape
    oh, ow = r
    wn = (self.w_fast if self.w_fast is not None else self.w).numel()
    alg.append(n * h * w * self.cin * 4 + n * oh * ow * self.cout * 4 + wn * 2)
    detail.append((f"{h}x{w}x{self.cin}->{self.cout} k{self.kh}x{self.kw} s{self.stride[0]} tn{self.tn}" + (" maxpool-in" if pooled_input else ""), n * h * w * self.cin * 4,
                   n * oh * ow * self.cout * 4, wn * 2))
    return r


SplitConv.__call__ = wrapped
trunk(x)
torch.cuda.synchronize()
one = list(alg)
SplitConv.__call__ = orig
for _ in range(2):
    trunk(x)
torch.cuda.synchronize()
out_dir = os.path.join(ROOT, "gpurun_out")
os.makedirs(out_dir, exist_ok=True)
json.dump({"per_launch": detail}, open(os.path.join(out_dir, "conv_traffic_layers.json"), "w"))
print(json.dumps({"per_launch": detail}))
print(json.dumps({"conv_launches_per_forward": len(one), "algorithmic_bytes_per_forward": sum(one),
                  "algorithmic_bytes_per_launch_avg": sum(one) / len(one)}))
