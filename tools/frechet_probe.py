#!/usr/bin/env python3
"""Phase times of the device Frechet distance at d = 2048 (HIP events inside the library), full-rank (N = 3000) and
rank-deficient (N = 1000) inputs.  TISE_SYTRD_TWO_LAUNCH=1 selects round 1's two-launch-per-column tridiagonalisation,
TISE_CHOL_PIVOTED=1 the pivoted Cholesky without trying the unpivoted fast path first."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tests import _cases  # noqa: E402
from tise_toolbox_amd import device  # noqa: E402

dev = torch.device("cuda", 0)
print("tridiagonalisation:", "two launches per column (round 1)" if os.environ.get("TISE_SYTRD_TWO_LAUNCH") else "one fused launch per column")
solver = device.FrechetSolver(2048, dev)
solver.set_profiling(True)
for kind, n1, n2 in (("fullrank", 3000, 2600), ("rankdef", 1000, 1000)):
    m1, s1, m2, s2 = (torch.as_tensor(a, device=dev) for a in _cases.frechet_case_2048(kind, n1, n2))
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        res = solver.distance(m1, s1, m2, s2)
        torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    ph = solver.phase_ms()
    print(f"{kind}: fid {res['fid']:.12f} rank {res['rank']}  wall {wall:.2f} ms  " + "  ".join(f"{k} {v:.2f}" for k, v in ph.items() if k != "rank"))
    solver.prefactor(s2)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    res2 = solver.distance_prefactored(m2, m1, s1)
    torch.cuda.synchronize(); wall = (time.perf_counter() - t0) * 1e3
    print(f"   prefactored (other side): fid {res2['fid']:.12f}  tail wall {wall:.2f} ms  prefactor {solver.prefactor_ms():.2f} ms")
