#!/usr/bin/env python3
"""Headline benchmark: images/sec through InceptionV3 + FID (+ IS*) on 256x256 images, MI355X.

    python bench.py --gpus 1 --steps 60 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A STEP is one pass of the hot path over one batch of `--batch` (default 500) synthetic uint8
256x256x3 images that are already resident in HBM:
    resize 256->299 (PIL-exact, csrc/resize.hip) -> InceptionV3 trunk + fc (PyTorch-ROCm, fp32)
    -> fp64 covariance/mean accumulation (csrc/stats.hip) -> IS* split sums (csrc/is_score.hip).
After the K steps the timed region also contains, once: the all-reduce of the sufficient
statistics over RCCL (world > 1), the finalisation of (mu, sigma), the Frechet distance against
pre-computed reference statistics (csrc/frechet.hip) and the IS* finalisation -- i.e. the whole job
"IS* + FID on K*batch images per GPU" (BASELINE.json configs[1] at the default K*batch = 30 000).
Each rank processes its own K*batch images (weak scaling); value = world * K * batch / seconds.

Prints ONE JSON line on rank 0 (contract in the task statement) with two extra objects:
  roofline      the dominant hand-written kernel of the step loop, timed with HIP events inside the
                timed region, against its gfx950 bound (see DESIGN.md "Measurement")
  cpu_baseline  the CPU oracle (oracle/, numpy/scipy/torch-CPU restatement of the reference) timed
                on this host on a bounded sample of the same workload (rank 0, N = 1 only)
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# gfx950 peaks.  HBM: /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW 8.0 TB/s spec".
# fp64 MFMA: the guide lists no f64 row; AMD's MI355X datasheet quotes 78.6 TFLOP/s FP64 matrix
# (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz), cross-checked by the v_mfma_f64_16x16x4_f64 issue-rate
# microbenchmark recorded in profiles/ (DESIGN.md "Peaks").
PEAK_HBM_GBS = 8000.0
PEAK_F64_MFMA_TFLOPS = 78.6
PEAK_F16_MFMA_TFLOPS = 2500.0    # MI355X_MICROARCH.md: "Peak BF16/FP16 MFMA ~2.5 PF dense"


def synth_images_device(lo, hi, device, seed=0, shift=0.0, hw=256):
    """Deterministic smooth 'MS-COCO-shaped' uint8 images for global indices [lo, hi), built on the
    GPU: per image and channel a sum of 4 random-orientation sinusoids and 4 gaussian blobs whose
    parameters are a hash of (seed, global index) -- identical for any sharding."""
    n = hi - lo
    idx = torch.arange(lo, hi, device=device, dtype=torch.float64)

    def u(j):   # uniform(0,1) hash of (index, stream j)
        x = torch.sin(idx * 12.9898 + (j + 1 + 131 * seed) * 78.233) * 43758.5453
        return (x - torch.floor(x)).float()

    yy, xx = torch.meshgrid(torch.linspace(0, 1, hw, device=device), torch.linspace(0, 1, hw, device=device), indexing="ij")
    out = torch.empty((n, hw, hw, 3), dtype=torch.uint8, device=device)
    j = 0
    for c in range(3):
        acc = torch.zeros((n, hw, hw), device=device)
        for _ in range(4):
            th = u(j) * 3.14159265; f = 1.0 + 11.0 * u(j + 1); ph = 6.2831853 * u(j + 2); a = 0.2 + 0.8 * u(j + 3)
            j += 4
            arg = 6.2831853 * f[:, None, None] * (torch.cos(th)[:, None, None] * xx + torch.sin(th)[:, None, None] * yy)
            acc += a[:, None, None] * torch.sin(arg + ph[:, None, None])
        for _ in range(4):
            cx, cy, s, a = u(j), u(j + 1), 0.03 + 0.27 * u(j + 2), 3.0 * u(j + 3) - 1.5
            j += 4
            acc += a[:, None, None] * torch.exp(-((xx - cx[:, None, None]) ** 2 + (yy - cy[:, None, None]) ** 2)
                                                / (2 * s[:, None, None] ** 2))
        mn = acc.amin(dim=(1, 2), keepdim=True)
        mx = acc.amax(dim=(1, 2), keepdim=True)
        img = (acc - mn) / (mx - mn + 1e-6) * (0.8 + shift) + 0.1 * u(j)[:, None, None]
        j += 1
        out[..., c] = (img.clamp(0, 1) * 255.0 + 0.5).to(torch.uint8)
    return out


def cpu_baseline(sample_u8, n_job, dims=2048):
    """Time the CPU oracle on a bounded sample (rank 0, N=1): resize + InceptionV3 forward on
    `sample` images, np.mean/np.cov on 3000x2048 fp64, the reference-form Frechet distance
    (scipy sqrtm) at d=2048, IS* reduction; combine to images/s for an n_job-image job."""
    from oracle import fid_oracle, inception_oracle, is_oracle, resize_oracle
    from tise_toolbox_amd.inception import build_inception3
    from tests import _cases
    threads = torch.get_num_threads()
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    n = sample_u8.shape[0]
    t0 = time.perf_counter()
    xs = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299)) for im in sample_u8])
    t_resize = time.perf_counter() - t0
    t0 = time.perf_counter()
    feats, logits = [], []
    for i in range(0, n, 16):
        o = inception_oracle.inception_forward(sd, torch.from_numpy(xs[i:i + 16]))[3]
        feats.append(o.flatten(1).numpy())
        logits.append(inception_oracle.logits_from_pool3(sd, o).numpy())
    t_fwd = time.perf_counter() - t0
    x = _cases.pool3_like_features(3000, dims, 42)
    y = _cases.pool3_like_features(2600, dims, 43, shift=0.1)
    t0 = time.perf_counter()
    m1, s1 = fid_oracle.calculate_activation_statistics(x)
    t_cov = time.perf_counter() - t0
    m2, s2 = fid_oracle.calculate_activation_statistics(y)
    t0 = time.perf_counter()
    fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    t_fd = time.perf_counter() - t0
    lg = np.random.default_rng(0).standard_normal((n_job, 1000)).astype(np.float32)
    t0 = time.perf_counter()
    is_oracle.inception_score_from_logits(lg, is_oracle.T_COCO, 10, "coco", dtype=np.float32)
    t_is = time.perf_counter() - t0
    per_img = (t_resize + t_fwd) / n + t_cov / 3000.0 + (t_fd + t_is) / n_job
    return {
        "value": 1.0 / per_img, "unit": "images/sec", "cores": threads, "kind": "port",
        "sample": (f"oracle/: PIL-exact resize + InceptionV3 fp32 (torch CPU, {threads} threads) on {n} images "
                   f"[{t_resize:.2f}s + {t_fwd:.2f}s], np.mean/np.cov on 3000x{dims} fp64 [{t_cov:.2f}s], reference-form "
                   f"Frechet distance (scipy sqrtm) d={dims} [{t_fd:.2f}s], IS* reduction {n_job}x1000 [{t_is:.2f}s]; "
                   f"per-image costs extrapolated linearly to a {n_job}-image job"),
        "host_cpus": os.cpu_count(),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=500)
    ap.add_argument("--ref-images", type=int, default=3000)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=48)
    ap.add_argument("--channels-last", type=int, default=-1)
    args = ap.parse_args()

    from tise_toolbox_amd import _lib, device, dist as tdist
    from tise_toolbox_amd.engine import RealismEngine, T_COCO, frechet_solver
    rank, world, local_rank = tdist.init_from_env()
    if world != args.gpus and rank == 0:
        print(f"[bench] WORLD_SIZE={world} but --gpus {args.gpus}; using WORLD_SIZE", file=sys.stderr)
    if os.environ.get("TISE_FORCE_DEVICE0"):       # testing aid: several ranks on one GPU (gloo backend only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cl = None if args.channels_last < 0 else bool(args.channels_last)
    eng = RealismEngine(dims=2048, device_index=local_rank, seed=0, with_logits=True, channels_last=cl)
    B, K, W = args.batch, args.steps, args.warmup
    n_rank = K * B
    n_total = n_rank * world
    lo = rank * n_rank

    # ---- inputs resident in HBM before the timed region ------------------------------------------
    data = torch.empty((n_rank, 256, 256, 3), dtype=torch.uint8, device=dev)
    for i in range(0, n_rank, 1000):
        j = min(i + 1000, n_rank)
        data[i:j] = synth_images_device(lo + i, lo + j, dev, seed=0)
    # reference statistics (the README recipe passes them as an .npz, fid_score.py:200-203): untimed
    eng.begin(n_total=args.ref_images)
    for i in range(0, args.ref_images, B):
        j = min(i + B, args.ref_images)
        eng.step_u8(synth_images_device(i, j, dev, seed=1, shift=0.12), i)
    mu_ref, sigma_ref = eng.statistics()
    solver = frechet_solver(2048, dev)
    solver.set_profiling(True)

    # ---- warmup (MIOpen solver search, allocator, clocks) -----------------------------------------
    eng.begin(n_total=max(W, 1) * B)
    for s in range(W):
        eng.step_u8(data[(s % K) * B:(s % K + 1) * B], s * B)
    if W > 0:
        eng.reduce()                                # also brings up the RCCL communicator outside the timed region
        mu_w, sig_w = eng.statistics()
        solver.distance(mu_w, sig_w, mu_ref, sigma_ref)
    elif world > 1:
        tdist.all_reduce_sum_(torch.zeros(1024, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()

    # ---- timed region -------------------------------------------------------------------------------
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(K)]
    from tise_toolbox_amd.conv_split import SplitConv as _SC
    _SC.timer = []                                  # HIP events around every convolution launch of the timed steps
    # The per-launch event pairs cost ~1.5 % of a step (130 extra records per 65 launches: measured 26.0 vs 25.6
    # ms/step), so they bracket the conv launches of every 6th timed step only (all steps when K < 12); the rocprofv3
    # summary of the same command is the cross-check.  TISE_BENCH_MODE=noevents | graph are experiment switches
    # (no conv events at all / hipGraph replay through RealismEngine.step_u8).
    mode = os.environ.get("TISE_BENCH_MODE", "events")
    conv_timer = [] if mode == "events" else None
    every = 6 if K >= 12 else 1
    _SC.timer = None
    eng.begin(n_total=n_total, temperature=T_COCO, splits=10, rule="coco")
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(K):
        batch = data[s * B:(s + 1) * B]
        if mode == "graph":
            eng.step_u8(batch, lo + s * B)
            continue
        _SC.timer = conv_timer if (conv_timer is not None and s % every == 0) else None
        ev[s][0].record()
        if getattr(eng, "_u8_stem", False):
            x = device.resize_u8_only(batch, (299, 299))       # uint8 out; the stem conv applies the input table
            ev[s][1].record()
            feats, logits = eng._trunk_u8(x)
        else:
            x = device.resize_bilinear_u8(batch, (299, 299), eng.lut, channels_last=eng.channels_last)
            ev[s][1].record()
            feats, logits = eng._trunk(x, prenormalized=True)
        ev[s][2].record()
        eng.stats.update_parts(feats, cov=True, col_sum=False)
        ev[s][3].record()
        eng.stats.update_parts(feats, cov=False, col_sum=True)
        eng.is_acc.update(logits, lo + s * B)
        if os.environ.get("TISE_BENCH_CHECKSUM"):      # debugging aid: per-batch feature / input checksums
            print(f"[chk] rank {rank} first_index {lo + s * B} feats {feats.double().sum().item()!r} "
                  f"imgs {batch.double().sum().item()!r} logits {logits.double().sum().item()!r}", file=sys.stderr, flush=True)
    _SC.timer = None
    t_loop_host = time.perf_counter()
    eng.reduce()                                             # RCCL all-reduce of {n, s, S} and the IS* sums
    mu, sigma = eng.statistics()
    res = solver.distance(mu, sigma, mu_ref, sigma_ref)      # one device->host read of 8 doubles
    is_mean, is_std = eng.inception_score()
    tdist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())

    from tise_toolbox_amd.trunk import SplitTrunk
    from tise_toolbox_amd.conv_split import SplitConv
    conv_events = conv_timer or []
    timed_steps = len(range(0, K, every)) if conv_timer is not None else 0
    SplitConv.timer = None
    if isinstance(eng.fused, SplitTrunk):
        # fp32-class arithmetic: every operand carried as two fp16 numbers (22 mantissa bits), three fp16 MFMAs
        # per product, fp32 accumulation; measured error vs an fp64 convolution is below MIOpen's fp32 kernels
        # (tools/conv_split_probe.py, tests/test_gpu_kernels.py::test_conv_split_matches_fp64_conv)
        conv_dtype = "f32 (split 2xf16 operands, 3 f16 MFMAs/product, f32 accumulate)"
        trunk_desc = "hand-written HIP implicit-GEMM convs (csrc/conv_split.hip) + HIP epilogues, BN folded, NHWC"
    else:
        conv_dtype = "f32"
        trunk_desc = "PyTorch-ROCm (MIOpen) fp32 convs + HIP epilogues, BN folded, " + ("channels_last" if eng.channels_last else "NCHW")
    if rank == 0:
        if mode == "graph":
            resize_ms = trunk_ms = syrk_ms = 1e-9
        else:
            resize_ms = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in range(K)]))
            trunk_ms = float(np.mean([ev[s][1].elapsed_time(ev[s][2]) for s in range(K)]))
            syrk_ms = float(np.mean([ev[s][2].elapsed_time(ev[s][3]) for s in range(K)]))
        phases = solver.phase_ms()
        d = 2048
        tiles = d // 64
        syrk_flop = 2.0 * B * 64 * 64 * (tiles * (tiles + 1) // 2)         # upper 64x64 tiles only, per launch
        resize_bytes = B * (256 * 256 * 3 + 299 * 299 * 3 * (1 if getattr(eng, '_u8_stem', False) else 4))
        kern = {
            "syrk_f32_upper_bk64_kernel": {"bound": "mfma", "achieved": syrk_flop / (syrk_ms * 1e-3) / 1e12,
                                      "peak": PEAK_F64_MFMA_TFLOPS, "unit": "TFLOP/s", "avg_ms": syrk_ms,
                                      "algorithmic_flop_per_launch": syrk_flop},
            "resize_bilinear_u8_kernel": {"bound": "hbm", "achieved": resize_bytes / (resize_ms * 1e-3) / 1e9,
                                          "peak": PEAK_HBM_GBS, "unit": "GB/s", "avg_ms": resize_ms,
                                          "algorithmic_bytes_per_launch": resize_bytes},
        }
        if conv_events:
            conv_ms = sum(a.elapsed_time(b) for a, b, _ in conv_events)
            conv_flop = sum(f for _, _, f in conv_events)
            n_launch = len(conv_events)
            # algorithmic work = the convolution's 2*M*N*K flop.  The kernel spends THREE fp16 MFMA flop per
            # algorithmic flop (hi*hi, hi*lo, lo*hi), so its ceiling is the dense fp16 MFMA peak / 3.
            kern["conv_split_fast_kernel"] = {
                "instances": "conv_split_fast_kernel<TN=1..5> (64 launches) + conv_win32_kernel (Conv2d_2a): all 65 conv launches of a step, as rocprofv3 lists them "
                             "(profiles/r01k_bench_steps10_kernel_stats.md)",
                "clock_note": "in-kernel stamps (profiles/r01g_conv_pipe_stamps.txt): 1.49 GHz while the MFMAs are busy, i.e. "
                              "~1550 TFLOP/s fp16 actually available; peak below is the 2.4 GHz datasheet figure / 3",
                "bound": "mfma", "achieved": conv_flop / (conv_ms * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
                "unit": "TFLOP/s", "avg_ms": conv_ms / timed_steps, "avg_launch_ms": conv_ms / n_launch,
                "launches_per_step": n_launch / timed_steps, "algorithmic_flop_per_step": conv_flop / timed_steps,
                "steps_with_events": timed_steps,
                "mfma_tflops_f16": 3.0 * conv_flop / (conv_ms * 1e-3) / 1e12, "mfma_peak_f16": PEAK_F16_MFMA_TFLOPS}
        for k in kern.values():
            k["frac"] = k["achieved"] / k["peak"]
        traffic = None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
        dom = max(kern, key=lambda k: kern[k]["avg_ms"])
        if os.path.exists(pmc_path):
            try:
                traffic = json.load(open(pmc_path)).get(dom, {}).get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": kern[dom]["bound"], "achieved": kern[dom]["achieved"],
                    "peak": kern[dom]["peak"], "unit": kern[dom]["unit"], "frac": kern[dom]["frac"],
                    "traffic": traffic, "avg_launch_ms": kern[dom].get("avg_launch_ms", kern[dom]["avg_ms"]),
                    "kernels": kern}
        out = {
            "metric": "images/sec through InceptionV3+FID on 30k 256x256 @1/2/4/8 GPU; |dFID| vs ref",
            "value": n_total / elapsed, "unit": "images/sec", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": conv_dtype, "data": "synthetic",
            "config": {"workload": f"IS*+FID on {n_rank} synthetic 256x256 images per GPU (BASELINE configs[1]: "
                                   f"30k images, 1xMI355X, InceptionV3 pool3 2048-d), batch {B}, seeded stand-in "
                                   f"InceptionV3 weights, reference stats from {args.ref_images} images",
                       "batch": B, "images_per_gpu": n_rank, "images_total": n_total, "dims": 2048,
                       "trunk": trunk_desc,
                       "parallelism": f"dp{world}"},
            "roofline": roofline,
            "stage_ms_per_step": {"resize": resize_ms, "trunk_fp32": trunk_ms, "cov_syrk": syrk_ms},
            "finalize_ms": {"host_wall_after_loop": (t1 - t_loop_host) * 1e3, **{k: v for k, v in phases.items()}},
            "scores": {"fid": float(res["fid"]), "is_mean": is_mean, "is_std": is_std, "rank": res["rank"],
                       "flags": res["flags"]},
            "trunk_tflops": 11.42e9 * B / (trunk_ms * 1e-3) / 1e12,
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = data[:args.cpu_sample].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(sample, n_rank)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    tdist.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
