#!/usr/bin/env python3
"""The constant of a tridiagonalisation launch: tise_eigvalsh on small symmetric matrices (n = 32 ... 512: a handful of
workgroups per launch, no bandwidth to speak of), time per column launch = the dependent-launch chain (kernel boundary +
the loads of what the previous launch wrote + two barriers + reductions)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd import device

dev = torch.device("cuda", 0)
solver = device.FrechetSolver(2048, dev)
solver.set_profiling(True)
g = torch.Generator(device="cpu").manual_seed(0)
for n in (32, 64, 128, 256, 512, 1024, 2048):
    a = torch.randn((n, n), generator=g, dtype=torch.float64)
    a = (a @ a.t()).to(dev)
    w_ref = torch.linalg.eigvalsh(a.cpu())
    for _ in range(3):
        w = solver.eigvalsh(a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        w = solver.eigvalsh(a)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    err = float((w.cpu().sort().values - w_ref).abs().max() / w_ref.abs().max())
    print(f"n = {n:5d}: eigvalsh {ms:8.3f} ms = {ms * 1e3 / max(1, n - 1):6.2f} us per column launch (incl. bisection)   max rel err {err:.1e}", flush=True)
