#!/usr/bin/env python3
"""Per-launch time of the split-precision convs inside one SplitTrunk forward (batch 500), and -- round 5 -- each layer's own
roofline: the larger of its matrix-core bound (3 fp16 MFMA flop per algorithmic flop at the 2.5 PF datasheet peak) and its HBM
bound (input tensor read once + output written once, 4 B per element in the split format; at the 8 TB/s spec and at the
~5 TB/s a streaming kernel reaches on this part).  Layers with K <= 288 are HBM-bound in this format, not matrix-core-bound."""
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
    nbytes = xs.numel() * 2 + sum((c1 - c0) * 4 * dst.shape[0] * dst.shape[1] * dst.shape[2] for c0, c1, dst, _, _ in segs)
    recs.append((e0, e1, f"{h}x{w}x{self.cin}->{self.cout} k{self.kh}x{self.kw} s{self.stride[0]} tn{self.tn} {kind}",
                 2.0 * n * oh * ow * self.cout * self.k, nbytes))
    return r


SplitConv.__call__ = wrapped
trunk(x)
torch.cuda.synchronize()
rows = [(a.elapsed_time(b), name, fl, nb) for a, b, name, fl, nb in recs]
tot = sum(r[0] for r in rows)
print(f"{len(rows)} conv launches, {tot:.2f} ms, {sum(r[2] for r in rows)/tot/1e9:.0f} TF-eq")
agg = {}
for ms, name, fl, nb in rows:
    a = agg.setdefault(name, [0.0, 0, 0.0, 0.0])
    a[0] += ms; a[1] += 1; a[2] += fl; a[3] += nb
floor8 = floor5 = 0.0
n_hbm, ms_hbm = 0, 0.0
for name, (ms, cnt, fl, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    t_mfma, t_hbm8, t_hbm5 = 3 * fl / 2.5e15 * 1e3, nb / 8e12 * 1e3, nb / 5e12 * 1e3
    floor8 += max(t_mfma, t_hbm8); floor5 += max(t_mfma, t_hbm5)
    hbm = t_hbm5 > t_mfma
    n_hbm += cnt if hbm else 0; ms_hbm += ms if hbm else 0.0
    print(f"{name:48s} x{cnt}  {ms:7.3f} ms ({100*ms/tot:4.1f}%)  {3*fl/ms/1e9:5.0f} TF16  {nb/ms/1e9:5.2f} TB/s   bounds: matrix cores {t_mfma:6.3f} ms, "
          f"HBM {t_hbm8:6.3f} (8 TB/s) / {t_hbm5:6.3f} (5 TB/s) -> {'HBM' if hbm else 'MFMA'}-bound, {ms / max(t_mfma, t_hbm5):4.2f} x its bound")
print(f"sum of the layers' own bounds: {floor8:.2f} ms (HBM at 8 TB/s) / {floor5:.2f} ms (HBM at 5 TB/s) against {tot:.2f} ms measured; "
      f"{n_hbm} launches ({ms_hbm:.2f} ms) are HBM-bound in the 4-byte split format")
