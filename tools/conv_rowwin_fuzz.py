#!/usr/bin/env python3
"""Random shapes through the split-precision convolution kernels against an fp64 convolution.
usage: conv_rowwin_fuzz.py [seed] [count] [variant = rowwin | fast | glds]   (rowwin: stride 1, KW >= 2)"""
import os
import random
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from tise_toolbox_amd.conv_split import SplitConv, merge, rowwin_fits, split  # noqa: E402


def run(seed=0, count=60, variant="rowwin", dev=None, verbose=True):
    dev = dev or torch.device("cuda:0")
    rnd = random.Random(seed)
    g = torch.Generator(device="cpu").manual_seed(seed + 5)
    bad, done = [], 0
    while done < count:
        kh = rnd.choice([1, 2, 3, 5, 7])
        kw = rnd.choice([2, 3, 4, 5, 7, 8] if variant == "rowwin" else [1, 2, 3, 5, 7])
        st = 1 if variant == "rowwin" else rnd.choice([1, 1, 2])
        ph, pw = rnd.randint(0, kh // 2 + 1), rnd.randint(0, kw // 2 + 1)
        H, W = rnd.randint(1, 24), rnd.randint(1, 40)
        if H + 2 * ph < kh or W + 2 * pw < kw:
            continue
        cin = rnd.choice([32, 48, 64, 80, 96, 112, 160])
        cout = rnd.choice([16, 48, 64, 80, 96, 144, 208])
        tn = rnd.choice([2, 3, 4] if variant == "rowwin" else [1, 2, 3, 4, 5])
        n = rnd.randint(1, 60)
        if variant == "rowwin" and not rowwin_fits(W + 2 * pw - kw + 1, kw):
            continue
        x = (torch.rand((n, H, W, cin), generator=g) * 2.0).to(dev)
        w = (torch.randn((cout, cin, kh, kw), generator=g) * (2.0 / (cin * kh * kw)) ** 0.5).to(dev)
        b = (torch.randn(cout, generator=g) * 0.2).to(dev)
        conv = SplitConv(w, b, (st, st), (ph, pw), dev, tn=tn, variant=variant)
        oh, ow = conv.out_hw(H, W)
        out = torch.full((n, oh, ow, 2 * cout), 9.0, dtype=torch.float16, device=dev)
        conv(split(x), [(0, cout, out, 0, 0)])
        ref = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), st, (ph, pw))).permute(0, 2, 3, 1)
        err = (merge(out).double() - ref).abs().max().item() / max(ref.abs().max().item(), 1e-30)
        done += 1
        if not err <= 4e-6:
            bad.append((dict(n=n, H=H, W=W, cin=cin, cout=cout, kh=kh, kw=kw, ph=ph, pw=pw, st=st, tn=tn), err))
            if verbose:
                print("MISMATCH", bad[-1], flush=True)
    return done, bad


if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    variant = sys.argv[3] if len(sys.argv) > 3 else "rowwin"
    done, bad = run(seed, count, variant)
    print(f"{done} random {variant} configurations, {len(bad)} mismatches")
