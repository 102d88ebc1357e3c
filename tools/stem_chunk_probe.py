#!/usr/bin/env python3
"""Stem phase (Conv2d_1a .. second max-pool) of the trunk at batch 500 in one go vs in chunks whose
intermediates (147^2 x 64 x 4 B = 5.5 MB per image) fit the 256 MB Infinity Cache."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.inception import InceptionV3
from tise_toolbox_amd.trunk import SplitTrunk, _lib, _p, _stream

dev = torch.device("cuda:0")
t = SplitTrunk(InceptionV3([3], seed=0), dev)
B = 500
x = torch.rand((B, 299, 299, 3), device=dev)


def stem(xc):
    n, h, w, _ = xc.shape
    oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    a = t._new(n, oh, ow, 32, xc.device)
    _lib.call("tise_stem_conv3x3s2_split", _p(xc), n, h, w, _p(t.stem_w), _p(t.c1a.b), _p(a), _stream())
    a = t._sconv(t.s2a, a)
    a = t._maxpool_split(t._sconv(t.s2b, a))
    a = t._sconv(t.s3b, a)
    return t._maxpool_split(t._sconv(t.s4a, a))


def timeit(fn, n=6):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


print(f"stem phase, one batch of {B}: {timeit(lambda: stem(x)):.2f} ms", flush=True)
for c in (250, 125, 100, 50, 25, 20, 10):
    xs = [x[i:i + c].contiguous() for i in range(0, B, c)]
    print(f"stem phase, {B // c} chunks of {c}: {timeit(lambda: [stem(xi) for xi in xs]):.2f} ms", flush=True)
