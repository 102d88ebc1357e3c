#!/usr/bin/env python3
"""Per-layer timing of the fused trunk's convolutions through MIOpen (find mode), batch 500, channels-last fp32."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("MIOPEN_DEBUG_CONV_DIRECT_NAIVE_CONV_FWD", "0")
from tise_toolbox_amd.inception import InceptionV3  # noqa: E402
from tise_toolbox_amd.trunk import FusedTrunk  # noqa: E402

torch.backends.cudnn.benchmark = True
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 500
m = InceptionV3([3], seed=0)
ft = FusedTrunk(m, dev)
records = []
orig = FusedTrunk._conv


def timed_conv(x, c):
    y = orig(x, c)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        y = orig(x, c)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    n, h, w, cin = x.shape
    cout, _, kh, kw = c.w.shape
    oh, ow = y.shape[1], y.shape[2]
    flop = 2.0 * n * oh * ow * cout * cin * kh * kw
    records.append({"in": [h, w, cin], "out": [oh, ow, cout], "k": [kh, kw], "s": list(c.stride), "ms": ms,
                    "gflop": flop / 1e9, "tflops": flop / ms / 1e9})
    return y


FusedTrunk._conv = staticmethod(timed_conv)
x = torch.rand((B, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
ft(x)
tot = sum(r["ms"] for r in records)
print(f"convs {len(records)} total {tot:.2f} ms  {sum(r['gflop'] for r in records)/tot:.1f} TF/s avg")
for r in sorted(records, key=lambda r: -r["ms"])[:70]:
    print(f"{str(r['in']):18s} -> {str(r['out']):18s} k{r['k']} s{r['s']}  {r['ms']:7.3f} ms  {r['tflops']:6.1f} TF  ({100*r['ms']/tot:4.1f}%)")
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(records, open(os.path.join(ROOT, "gpurun_out", "conv_probe.json"), "w"))
