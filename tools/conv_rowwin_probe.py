#!/usr/bin/env python3
"""Row-window kernel against the default kernel on the layers it applies to (stride 1, KW > 1):
error of both against an fp64 convolution, repeatability, time per launch (batch from argv, default 500)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.conv_split import SplitConv, merge, split
dev = torch.device("cuda:0")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
g = torch.Generator(device="cpu").manual_seed(1)
layers = [(17, 192, 192, 1, 7, (0, 3), 3), (17, 160, 160, 1, 7, (0, 3), 5), (17, 160, 192, 1, 7, (0, 3), 3), (17, 128, 128, 1, 7, (0, 3), 4),
          (17, 128, 192, 1, 7, (0, 3), 3), (35, 96, 96, 3, 3, (1, 1), 3), (35, 64, 96, 3, 3, (1, 1), 3), (8, 448, 384, 3, 3, (1, 1), 4),
          (8, 384, 384, 1, 3, (0, 1), 4), (147, 32, 64, 3, 3, (1, 1), 2), (73, 80, 192, 3, 3, (0, 0), 3), (35, 48, 64, 5, 5, (2, 2), 2)]


def timed(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


tf = tw = 0.0
for (H, Cin, Cout, kh, kw, pad, tn) in layers:
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    x32 = torch.relu(torch.randn((N, H, H, Cin), device=dev))
    x = split(x32)
    nref = min(N, 3)
    ref64 = torch.relu(torch.conv2d(x32[:nref].permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, pad)).permute(0, 2, 3, 1)
    scale = ref64.abs().max().item()
    cf = SplitConv(w, b, (1, 1), pad, dev, tn=tn, variant="fast")
    oh, ow = cf.out_hw(H, H)
    out = torch.zeros((N, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
    cf(x, [(0, Cout, out, 0, 0)])
    ef = (merge(out)[:nref].double() - ref64).abs().max().item() / scale
    ms_f = timed(lambda: cf(x, [(0, Cout, out, 0, 0)]))
    res = []
    for tnw in sorted({min(max(tn, 2), 4), 3} if Cout > 64 else {2}):
        cw = SplitConv(w, b, (1, 1), pad, dev, tn=tnw, variant="rowwin")
        o2 = torch.full_like(out, 3.0)
        cw(x, [(0, Cout, o2, 0, 0)])
        first = o2.clone()
        same = True
        for rep in range(3):
            o2.fill_(3.0)
            cw(x, [(0, Cout, o2, 0, 0)])
            same = same and bool(torch.equal(o2, first))
        ew = (merge(o2)[:nref].double() - ref64).abs().max().item() / scale
        res.append((tnw, timed(lambda: cw(x, [(0, Cout, o2, 0, 0)])), ew, same))
    best = min(r[1] for r in res)
    tf += ms_f; tw += min(best, ms_f)
    print(f"{H}x{H}x{Cin}->{Cout} k{kh}x{kw}: fast tn{tn} {ms_f:.3f} ms (err {ef:.1e}) | rowwin " +
          "  ".join(f"tn{t} {m:.3f} ({ms_f / m:.2f}x, err {e:.1e}, repeatable={sm})" for t, m, e, sm in res), flush=True)
print(f"sum fast {tf:.2f} ms, best-of {tw:.2f} ms")
