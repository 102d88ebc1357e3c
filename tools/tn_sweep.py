#!/usr/bin/env python3
"""Every distinct trunk conv shape x tile width TN = 1..5 with the default kernel: which TN is fastest (batch 500)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, split, pick_tn
from tise_toolbox_amd.inception import InceptionV3
from tise_toolbox_amd.trunk import SplitTrunk


def trunk_layers():
    """(H, Cin, Cout, kh, kw, stride, pad, count) of every distinct conv launch of a SplitTrunk forward."""
    seen = {}
    trunk = SplitTrunk(InceptionV3([3], seed=0), torch.device("cuda:0"))
    orig = SplitConv.__call__

    def rec(self, xs, segs):
        key = (xs.shape[1], self.cin, self.cout, self.kh, self.kw, self.stride[0], self.padding)
        seen[key] = seen.get(key, 0) + 1
        return orig(self, xs, segs)
    SplitConv.__call__ = rec
    trunk(torch.rand((2, 3, 299, 299), device="cuda:0").contiguous(memory_format=torch.channels_last))
    SplitConv.__call__ = orig
    return [k + (c,) for k, c in seen.items()]


LAYERS = trunk_layers()

dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
tot_cur = tot_best = 0.0
for (H, Cin, Cout, kh, kw, st, pad, cnt) in LAYERS:
    g = torch.Generator(device="cpu").manual_seed(1)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    x = split(torch.relu(torch.randn((N, H, H, Cin), device=dev)))        # ReLU-like sparsity, as inside the trunk
    res = {}
    for tn in range(1, 6):
        if 32 * tn >= 2 * max(32, Cout) and tn > 1:
            continue
        conv = SplitConv(w, b, (st, st), pad, dev, tn=tn, variant="fast")
        oh, ow = conv.out_hw(H, H)
        out = torch.zeros((N, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
        for _ in range(3):
            conv(x, [(0, Cout, out, 0, 0)])
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            conv(x, [(0, Cout, out, 0, 0)])
        e1.record(); torch.cuda.synchronize()
        res[tn] = e0.elapsed_time(e1) / 10
        del out
    cur = pick_tn(Cout)
    best = min(res, key=res.get)
    tot_cur += res[cur] * cnt; tot_best += res[best] * cnt
    print(f"{H}x{H}x{Cin}->{Cout} k{kh}x{kw} s{st} x{cnt}: cur tn{cur} {res[cur]:.3f}  best tn{best} {res[best]:.3f}  " +
          " ".join(f"tn{t}={v:.3f}" for t, v in res.items()), flush=True)
    del x
print(f"total: current table {tot_cur:.2f} ms, per-layer best {tot_best:.2f} ms")
