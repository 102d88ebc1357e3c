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


# ---- round 5: the ACCUMULATE side at 80 classes -------------------------------------------------------
# device work only (features already computed): (a) the plain single-accumulator update per device batch (what the plain FID
# loop does), (b) round 4's per-class loop -- one index_select + one covariance launch per class present in a device batch --
# (c) round 5: rows kept, ONE class sort + ONE grouped launch per directory.
def _accumulate_probe(n_rows=24000, batch=3000, ncls=80):
    g = torch.Generator(device="cpu").manual_seed(0)
    feats = torch.rand((n_rows, 2048), generator=g).to(dev)
    cls = torch.randint(0, ncls, (n_rows,), generator=g)
    torch.cuda.synchronize()

    def timed(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / reps
    plain = device.StatsAccumulator(2048, dev)

    def run_plain():
        for a in range(0, n_rows, batch):
            plain.update(feats[a:a + batch])
    accs = [device.StatsAccumulator(2048, dev) for _ in range(ncls)]

    def run_r4():
        for a in range(0, n_rows, batch):
            f, c = feats[a:a + batch], cls[a:a + batch]
            for k in torch.unique(c).tolist():
                idx = torch.nonzero(c == k).flatten().to(dev)
                accs[k].update(f.index_select(0, idx))

    def run_r5():
        order = torch.argsort(cls, stable=True)
        counts = torch.bincount(cls, minlength=ncls)
        offsets = [0] + torch.cumsum(counts, 0).tolist()
        device.stats_update_grouped(accs, feats.index_select(0, order.to(dev)), offsets)
    t_plain, t_r4, t_r5 = timed(run_plain), timed(run_r4), timed(run_r5)
    print(f"accumulate side, {n_rows} feature rows, {ncls} classes, device batches of {batch}: plain single accumulator {t_plain * 1e3:.2f} ms; "
          f"round 4 per-class loop {t_r4 * 1e3:.1f} ms; round 5 one sort + ONE grouped launch {t_r5 * 1e3:.2f} ms "
          f"(a 3000-image trunk pass is ~112 ms: {n_rows // batch} passes = {n_rows // batch * 112} ms)", flush=True)


_accumulate_probe()
