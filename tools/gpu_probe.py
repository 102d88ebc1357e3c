#!/usr/bin/env python3
"""Development probe (GPU box): trunk throughput by layout/batch, hand-kernel timings, Frechet phases.
Writes gpurun_out/probe.json.  Not part of the product path."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd import device  # noqa: E402
from tise_toolbox_amd.engine import RealismEngine, frechet_solver  # noqa: E402
from tests import _cases  # noqa: E402

dev = torch.device("cuda", 0)
out = {"device": torch.cuda.get_device_name(0)}


def timeit(fn, iters=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


which = sys.argv[1:] or ["trunk", "kernels", "frechet"]

if "trunk" in which:
    res = {}
    for cl in (True, False):
        eng = RealismEngine(dims=2048, seed=0, with_logits=True, channels_last=cl)
        for bs in (50, 250, 500):
            x = torch.rand((bs, 3, 299, 299), device=dev)
            if cl:
                x = x.contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                ms = timeit(lambda: eng._trunk(x, True), iters=4, warm=3)
            res[f"{'nhwc' if cl else 'nchw'}_b{bs}"] = {"ms": ms, "img_s": bs / ms * 1e3, "tflops": 11.42e9 * bs / ms / 1e9}
            print("trunk", cl, bs, res[f"{'nhwc' if cl else 'nchw'}_b{bs}"], flush=True)
        del eng
        torch.cuda.empty_cache()
    out["trunk"] = res

if "fused" in which:
    res = {}
    for fused in ("split", "miopen", False):
        if fused:
            os.environ["TISE_CONV"] = fused
        eng = RealismEngine(dims=2048, seed=0, with_logits=True, channels_last=True, fused=bool(fused))
        for bs in (500,):
            x = torch.rand((bs, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
            with torch.no_grad():
                ms = timeit(lambda: eng._trunk(x, True), iters=5, warm=3)
            res[f"{fused or 'module'}_b{bs}"] = {"ms": ms, "img_s": bs / ms * 1e3, "tflops": 11.42e9 * bs / ms / 1e9}
            print("trunk", fused, bs, res[f"{fused or 'module'}_b{bs}"], flush=True)
        del eng
        torch.cuda.empty_cache()
    out["fused"] = res

if "kernels" in which:
    res = {}
    imgs = torch.randint(0, 256, (500, 256, 256, 3), dtype=torch.uint8, device=dev)
    lut = device.make_lut(True)
    for cl in (True, False):
        ms = timeit(lambda: device.resize_bilinear_u8(imgs, (299, 299), lut, channels_last=cl), iters=10)
        res[f"resize_b500_{'nhwc' if cl else 'nchw'}"] = {"ms": ms, "GBs": 500 * 1269420 / ms / 1e6}
    for rows in (50, 500, 3000):
        f = torch.rand((rows, 2048), device=dev)
        acc = device.StatsAccumulator(2048, dev)
        ms = timeit(lambda: acc.update_parts(f, True, False), iters=10)
        flop = 2.0 * rows * 64 * 64 * 528
        res[f"syrk_rows{rows}"] = {"ms": ms, "tflops": flop / ms / 1e9}
        ms2 = timeit(lambda: acc.update_parts(f, False, True), iters=10)
        res[f"colsum_rows{rows}"] = {"ms": ms2}
    a = torch.randn((2048, 2048), dtype=torch.float64, device=dev)
    ms = timeit(lambda: device.gemm_f64(a, a), iters=5)
    res["gemm_f64_2048"] = {"ms": ms, "tflops": 2 * 2048 ** 3 / ms / 1e9}
    lg = torch.randn((500, 1000), device=dev)
    isa = device.InceptionScoreAccumulator(1000, 30000, 0.9, 10, "coco", False, dev)
    res["is_update_b500"] = {"ms": timeit(lambda: isa.update(lg, 0), iters=10)}
    out["kernels"] = res
    print(json.dumps(res, indent=1), flush=True)

if "frechet" in which:
    res = {}
    from oracle import fid_oracle
    for kind, n1, n2 in (("fullrank", 3000, 2600), ("rankdef", 1000, 1000)):
        mu1, s1, mu2, s2 = _cases.frechet_case_2048(kind, n1, n2)
        solver = frechet_solver(2048, dev)
        solver.set_profiling(True)
        t = [torch.as_tensor(v, device=dev) for v in (mu1, s1, mu2, s2)]
        solver.distance(*t)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = solver.distance(*t)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        ph = solver.phase_ms()
        g = np.load(os.path.join(ROOT, "tests", "golden", f"frechet_d2048_{kind}.npz"))
        res[kind] = {"wall_ms": wall * 1e3, "phases": ph, "fid": float(r["fid"]), "ref": float(g["fid"]),
                     "diff": float(r["fid"]) - float(g["fid"]), "rank": r["rank"], "neg": r["n_negative"]}
        print(kind, res[kind], flush=True)
    out["frechet"] = res

os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "probe_" + "_".join(which) + ".json"), "w") as f:
    json.dump(out, f, indent=1)
print("PROBE DONE")
