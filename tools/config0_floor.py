#!/usr/bin/env python3
"""BASELINE configs[0] (1 000 + 1 000 white-noise 256x256 images, batch 50): where does |dFID| come from?

Prints, on the SAME images and stand-in weights:
  FID of the CPU oracle with its convolutions on 16 host threads            (the number the tests compare against)
  FID of the CPU oracle with its convolutions on 1 host thread              (fp32-vs-fp32 floor: only oneDNN's
                                                                             summation order differs)
  FID of the device path, split-fp16 HIP trunk                              (the product)
  FID of the device path, exact-fp32 MIOpen trunk (TISE_CONV=miopen)
and the three distances VERDICT r2 asked for.  The 1-thread oracle runs as 16 single-threaded worker processes, each
on its own run of whole batches: per-batch arithmetic is independent, so the result is that of one 1-thread process.

    python tools/config0_floor.py [n_images_per_side]
"""
import multiprocessing as mp
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BS = 50


def images(seed, n):
    return np.random.default_rng(seed).integers(0, 256, (n, 256, 256, 3), dtype=np.uint8)


def _oracle_rows(args):
    seed, n, lo, hi, threads = args
    import torch
    from oracle import inception_oracle, resize_oracle
    from tise_toolbox_amd.inception import build_inception3
    torch.set_num_threads(threads)
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    torch.set_num_threads(threads)
    u8 = images(seed, n)[lo:hi]
    out = []
    for i in range(0, hi - lo, BS):
        x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299)) for im in u8[i:i + BS]])
        out.append(inception_oracle.inception_forward(sd, torch.from_numpy(x))[3].flatten(1).numpy())
    return np.concatenate(out)


def oracle_fid(n, threads, procs):
    from oracle import fid_oracle
    stats = []
    for seed in (1, 0):                                   # path1 = ref (seed 1), path2 = gen (seed 0)
        nb = n // BS
        cuts = [(nb * p // procs) * BS for p in range(procs + 1)]
        jobs = [(seed, n, cuts[p], cuts[p + 1], threads) for p in range(procs) if cuts[p + 1] > cuts[p]]
        if procs == 1:
            rows = [_oracle_rows(jobs[0])]
        else:
            with mp.get_context("spawn").Pool(len(jobs)) as pool:
                rows = pool.map(_oracle_rows, jobs)
        stats.append(fid_oracle.calculate_activation_statistics(np.concatenate(rows).astype(np.float64)))
    return float(fid_oracle.calculate_frechet_distance(*stats[0], *stats[1]))


def device_fid(n, conv):
    import torch
    from tise_toolbox_amd import device
    from tise_toolbox_amd.engine import RealismEngine, frechet_solver
    os.environ["TISE_CONV"] = conv
    os.environ["TISE_MIOPEN_FIND"] = "0"
    dev = torch.device("cuda", 0)
    eng = RealismEngine(dims=2048, seed=0)
    stats = []
    for seed in (1, 0):
        u8 = images(seed, n)
        eng.begin()
        for i in range(0, n, BS):
            eng.step_u8(torch.from_numpy(u8[i:i + BS]).to(dev), i)
        mu, sigma = eng.statistics()
        stats.append((mu.clone(), sigma.clone()))
    return float(frechet_solver(2048, dev).distance(*stats[0], *stats[1])["fid"])


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
    t = time.time()
    f16 = oracle_fid(n, 16, 1)
    t16 = time.time() - t
    t = time.time()
    f1 = oracle_fid(n, 1, 16)
    t1 = time.time() - t
    fs = device_fid(n, "split")
    fm = device_fid(n, "miopen")
    print(f"configs[0] at {n} + {n} images, batch {BS}, stand-in weights seed 0")
    print(f"  oracle, 16 threads            FID {f16:.9f}   ({t16:.0f} s)")
    print(f"  oracle, 1 thread              FID {f1:.9f}   ({t1:.0f} s, 16 single-threaded workers)")
    print(f"  device, split-fp16 HIP trunk  FID {fs:.9f}")
    print(f"  device, MIOpen fp32 trunk     FID {fm:.9f}")
    print(f"  |split  - oracle16| = {abs(fs - f16):.3e}")
    print(f"  |miopen - oracle16| = {abs(fm - f16):.3e}")
    print(f"  |oracle1 - oracle16| = {abs(f1 - f16):.3e}")
    print(f"  |split  - miopen|   = {abs(fs - fm):.3e}")


if __name__ == "__main__":
    main()
