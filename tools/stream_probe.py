#!/usr/bin/env python3
"""Does running two half-batches on two HIP streams hide launch tails / HBM-bound kernels behind the convs?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd.inception import InceptionV3
from tise_toolbox_amd.trunk import SplitTrunk

dev = torch.device("cuda:0")
m = InceptionV3([3], seed=0)
B = 500
x = torch.rand((B, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)


def timeit(fn, n=8):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


t_full = SplitTrunk(m, dev)
print(f"1 stream, batch {B}: {timeit(lambda: t_full(x)):.2f} ms per {B} images", flush=True)
for parts in (2, 4):
    trunks = [SplitTrunk(m, dev) for _ in range(parts)]
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs = [x[i * (B // parts):(i + 1) * (B // parts)].contiguous(memory_format=torch.channels_last) for i in range(parts)]

    def run():
        for t, s, xi in zip(trunks, streams, xs):
            with torch.cuda.stream(s):
                t(xi)
    print(f"{parts} streams, batch {B // parts} each: {timeit(run):.2f} ms per {B} images", flush=True)
    del trunks
for b2 in (250, 1000):
    xb = torch.rand((b2, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    tb = SplitTrunk(m, dev)
    print(f"1 stream, batch {b2}: {timeit(lambda: tb(xb)) * B / b2:.2f} ms per {B} images", flush=True)
    del tb, xb
