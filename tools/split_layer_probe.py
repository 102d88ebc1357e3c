#!/usr/bin/env python3
"""Per-launch time of the split-precision convs inside one SplitTrunk forward (batch 500)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd.conv_split import SplitConv  # noqa: E402
from tise_toolbox_amd.inception import InceptionV3  # noqa: E402
from tise_toolbox_amd.trunk import SplitTrunk  # noqa: E402

dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
m = InceptionV3([3], seed=0)
trunk = SplitTrunk(m, dev)
x = torch.rand((B, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
for _ in range(2):
    trunk(x)
orig = SplitConv.__call__
recs = []


def wrapped(self, xs, segs, pooled_input=False, **kw):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    r = orig(self, xs, segs, pooled_input=pooled_input, **kw)
    e1.record()
    n, h, w, _ = xs.shape
    oh, ow = r if pooled_input else self.out_hw(h, w)                # the convolution's own output grid (pool_output returns the pooled one)
    kind = "maxpool-in" if pooled_input else (f"pipe{self.pipe_cfg}" if self.pipe_cfg is not None else self.variant)
    if kw.get("pool_output"):
        kind += " maxpool-out"
    recs.append((e0, e1, f"{h}x{w}x{self.cin}->{self.cout} k{self.kh}x{self.kw} s{self.stride[0]} tn{self.tn} {kind}",
                 2.0 * n * oh * ow * self.cout * self.k))
    return r


SplitConv.__call__ = wrapped
trunk(x)
torch.cuda.synchronize()
rows = [(a.elapsed_time(b), name, fl) for a, b, name, fl in recs]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} conv launches, {tot:.2f} ms, {sum(r[2] for r in rows)/tot/1e9:.0f} TF-eq")
agg = {}
for ms, name, fl in rows:
    a = agg.setdefault(name, [0.0, 0, 0.0])
    a[0] += ms; a[1] += 1; a[2] += fl
for name, (ms, cnt, fl) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:48s} x{cnt}  {ms:7.3f} ms ({100*ms/tot:4.1f}%)  {3*fl/ms/1e9:5.0f} TF16")
