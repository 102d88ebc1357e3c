#!/usr/bin/env python3
"""Tail of `fid_score --per-class` (BASELINE configs[4]: 80 per-class O-FIDs): the 80 Frechet solves of one rank's share,
on 1 .. 4 streams (fid_score._solve_classes: one host thread + one solver + one stream each; a solve is ~4 000 dependent
launches, so concurrent solves interleave on the device).  Statistics: 80 classes x (rows_a, rows_b) pool3-like feature
rows -- few hundred crops per class as in a COCO crop directory, i.e. rank-deficient covariances (the diagonally pivoted
Cholesky path) -- and a full-rank case.  Usage: python tools/perclass_probe.py [classes] [rows]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from tests import _cases
from tise_toolbox_amd import device, fid_score

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
ncls = int(sys.argv[1]) if len(sys.argv) > 1 else 80
for rows in ([int(sys.argv[2])] if len(sys.argv) > 2 else [300, 2500]):
    pairs = []
    for c in range(ncls):
        a, b = device.StatsAccumulator(2048, dev), device.StatsAccumulator(2048, dev)
        a.update(torch.from_numpy(_cases.pool3_like_features(rows, 2048, 100 + c)).to(dev))
        b.update(torch.from_numpy(_cases.pool3_like_features(rows + 7, 2048, 500 + c, shift=0.1)).to(dev))
        pairs.append((a, b))
    torch.cuda.synchronize()
    base = None
    for nthr in (1, 2, 3, 4):
        os.environ["TISE_PERCLASS_STREAMS"] = str(nthr)
        fid_score._solve_classes(pairs[:2], 2048, dev)            # warm-up (solver handles, code objects)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fid_score._solve_classes(pairs, 2048, dev)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        base = base or out
        same = max(abs(x - y) for x, y in zip(out, base))
        print(f"{ncls} classes, {rows} / {rows + 7} rows per side, {nthr} stream(s): {dt * 1e3:8.1f} ms = {dt / ncls * 1e3:6.2f} ms per class "
              f"(max |difference| to 1 stream {same:.1e}); on 8 ranks: {dt / 8 * 1e3:6.1f} ms per rank", flush=True)
