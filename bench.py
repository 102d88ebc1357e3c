#!/usr/bin/env python3
"""Headline benchmark: images/sec through InceptionV3 + FID (+ IS*) on 30k 256x256 images, MI355X.

    python bench.py --gpus 1 --steps 60 --warmup 3
    python bench.py --gpus 8                      (no torchrun environment: starts the 8 ranks itself, see _self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

The JOB is BASELINE.json's metric and does not depend on the flags: IS* + FID of ONE set of 30 000 synthetic uint8
256x256x3 images -- decoded pixels already resident in HBM (PNG decode and the host->device copy are NOT in the timed
region; DESIGN.md section 6 gives the host-inclusive rates of the CLI) -- against pre-computed reference statistics.
A STEP is 1/K of the job = the hot path over 30 000 / K images (500 at the default K = 60, 1 500 at --steps 20; the
device runs them in batches of up to 5000, engine.DEVICE_BATCH_DEFAULT; the feed legs and the CLIs: 1000, engine.FEED_DEVICE_BATCH_DEFAULT; `--batch B` makes the job K*B images instead):
    resize 256->299 (PIL-exact, csrc/resize.hip) -> InceptionV3 trunk (hand-written split-fp16 MFMA convolutions,
    csrc/conv_split.hip / conv_pipe.hip / trunk_ops.hip) + fc -> fp64 covariance/mean accumulation (csrc/stats.hip)
    -> IS* split sums (csrc/is_score.hip).
With N GPUs the SAME job is sharded (STRONG scaling, SURVEY.md section 8(d) "Config 3"): rank r takes the
contiguous index range dist.shard_range(30 000, r, N) (3 750 images at N = 8) and runs it in device batches that
divide its range (3000 / 3000 / 2500 / 1875 images at N = 1 / 2 / 4 / 8; a step stays 1/K of the job).  After the loop the timed region contains, once: the
all-reduce of the sufficient statistics over RCCL, the finalisation of (mu, sigma), the Frechet distance
(csrc/frechet.hip, solved redundantly on every rank) and the IS* finalisation.  value = 30 000 / max-over-ranks
seconds; `allreduce_ms` and `finalize_ms` are reported separately.  `--scaling weak` gives every rank its own
30 000 images instead.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects:
  roofline      the dominant hand-written kernel of the step loop, timed with HIP events inside the timed region,
                against its gfx950 bound (DESIGN.md "Measurement")
  parity        |dFID|, |dIS| of the device path against the CPU oracle on the first images of the timed set
                (outside the timed region; N = 1 only)
  cross_check   the whole timed set again through exact-fp32 MIOpen convolutions: |dFID|, |dIS|, max feature error
                of the split-fp16 trunk at the job's full size (N = 1 only)
  cpu_baseline  the CPU oracle (oracle/: numpy/scipy/torch-CPU restatement of the reference) timed on this host on
                a bounded sample of the same workload (rank 0, N = 1 only)
  host_feed     the same job fed from page-locked HOST memory in 50-image slices (N = 1, never `value`)
  png_feed      the same job from 30 000 PNG FILES through the CLIs' feed: native inflate-only decode processes -> shared
                page-locked ring -> H2D -> PNG row filters on the GPU (csrc/png_unfilter.hip); `ratio_to_resident` (N = 1, never `value`)
  cli_process   the README recipe `python -m tise_toolbox_amd.fid_score --path1 ref.npz --path2 <those files>` as a FRESH CHILD
                process, twice: wall clock + TISE_TIMING phases (N = 1, never `value`)
  ranks         per-rank {loop_device_s, loop_host_enqueue_s, reduce_host_s, reduce_device_s, first_collective_s, total_s},
                all-gathered: a straggler or a slow communicator can be told from a slow trunk (every N)
"""
import argparse
import json
import os
import statistics
import sys
import tempfile
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
# what the bare split-precision inner loop sustains under the chip's power limit (16x16x32, 32 px x 128 couts per wave, two
# waves per SIMD; profiles/r03k_mfma_shape_probe.txt): reported next to the roofline, never used as its peak
SUSTAINED_F16_MFMA_TFLOPS = 1585.0
JOB_IMAGES = 30000         # BASELINE.json metric / configs[1]: 30k images (README.md:214-219 of the reference)
from tise_toolbox_amd.engine import DEVICE_BATCH_DEFAULT as DEVICE_BATCH   # 5000 for the resident job (the fed legs: engine.device_batch_images, 1000)
                                                                            # (a rank's share is cut into equal batches of at most this)


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


def rank_batch(n_rank, preferred=DEVICE_BATCH, cap=DEVICE_BATCH):
    """Device batch of a rank: the largest divisor of the rank's image count that is <= cap (30 000 / 15 000 / 7 500 / 3 750
    images at 1 / 2 / 4 / 8 GPUs -> 5000 / 5000 / 3750 / 3750); a count without a usable divisor runs `preferred` with a
    short tail."""
    if n_rank <= 0:
        return preferred
    if n_rank <= cap:
        return n_rank
    for b in range(min(cap, n_rank), 127, -1):
        if n_rank % b == 0:
            return b
    return preferred


def _median_time(fn, repeats=3, warmup=1, budget_s=None):
    """median wall time of fn() over `repeats` runs after `warmup` untimed runs; stops repeating once `budget_s`
    seconds have been spent (at least one timed run)."""
    for _ in range(warmup):
        fn()
    ts, spent = [], 0.0
    for _ in range(repeats):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
        spent += ts[-1]
        if budget_s is not None and spent > budget_s:
            break
    return statistics.median(ts), len(ts)


def cpu_reference_pass(gen_u8, ref_u8, n_job, dims=2048, workers=8):
    """The CPU oracle on THIS host (rank 0, N = 1), doing what the reference's CPU path does, stage by stage:

      decode+resize  PNG files -> PIL decode -> PIL bilinear 299x299 -> ToTensor, DataLoader(batch 50, 8 workers) as
                     fid_score.py:208-217 (Pillow is what the reference itself uses; resize_oracle is its restatement)
      forward        oracle/inception_oracle.py fp32 InceptionV3 (torch CPU, all threads) at batch 50
      cov            np.mean / np.cov(rowvar=False) on (n_job, 2048) float64            (fid_score.py:194-195)
      frechet        scipy.linalg.sqrtm form, d = 2048                                   (fid_score.py:121-171)
      is             the IS* reduction on (n_job, 1000) fp32 logits                      (inception_score_star_coco.py:52-60)

    1 warm-up + 3 repeats (median) for the per-image stages on len(gen_u8) images; cov / sqrtm once at the job's
    size.  Returns (cpu_baseline dict, oracle features/logits of gen and ref for the parity figures)."""
    import torch.utils.data as tud
    from PIL import Image
    from oracle import fid_oracle, inception_oracle, is_oracle, resize_oracle
    from tise_toolbox_amd.inception import build_inception3
    from tests import _cases
    # torch's CPU convolutions are fastest well below the host's thread count (this pool's 128-thread hosts:
    # 94 / 52 / 35 / 30 ms per image at 128 / 64 / 32 / 16 threads, batch 50): the baseline gets the better setting
    from tise_toolbox_amd.hostinfo import usable_cpus
    all_threads = torch.get_num_threads()
    threads = min(32, all_threads, usable_cpus())            # the cgroup's CPU quota counts too (16 CPUs on the GPU boxes)
    torch.set_num_threads(threads)
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    n = gen_u8.shape[0]
    bs = 50

    # ---- stage 1: PNG decode + PIL resize + ToTensor on 8 DataLoader workers (the reference's loader) ----
    class _Files(tud.Dataset):
        def __init__(self, files):
            self.files = files

        def __len__(self):
            return len(self.files)

        def __getitem__(self, i):
            img = Image.open(self.files[i]).convert("RGB").resize((299, 299), Image.BILINEAR)
            return torch.from_numpy(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1).copy()).float().div_(255.0)

    with tempfile.TemporaryDirectory(prefix="tise_bench_png_") as tmp:
        files = []
        for i in range(n):
            p = os.path.join(tmp, f"{i:05d}.png")
            Image.fromarray(gen_u8[i]).save(p)
            files.append(p)
        loader = tud.DataLoader(_Files(files), batch_size=bs, shuffle=False, drop_last=True, num_workers=workers)

        def decode_pass():
            for _ in loader:
                pass
        t_dec, r_dec = _median_time(decode_pass, repeats=3, warmup=1, budget_s=20.0)

    # ---- stage 2: CPU fp32 forward (batch 50) ----
    def to_input(u8):
        return np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299)) for im in u8])

    x_gen = to_input(gen_u8)
    out = {}

    def forward_pass(x=x_gen, key="gen"):
        feats, logits = [], []
        for i in range(0, x.shape[0], bs):
            o = inception_oracle.inception_forward(sd, torch.from_numpy(x[i:i + bs]))[3]
            feats.append(o.flatten(1).numpy())
            logits.append(inception_oracle.logits_from_pool3(sd, o, bias=False).numpy())   # IS* coco head: pool3 x W, no bias (:104-105)
        out[key] = (np.concatenate(feats), np.concatenate(logits))

    inception_oracle.inception_forward(sd, torch.from_numpy(x_gen[:bs]))          # warm-up: oneDNN primitive creation
    t_fwd, r_fwd = _median_time(forward_pass, repeats=3, warmup=0, budget_s=45.0)
    forward_pass(to_input(ref_u8), "ref")                                         # parity partner set (untimed)

    # ---- stages 3-5 at the job's size ----
    x = _cases.pool3_like_features(n_job, dims, 42)
    t0 = time.perf_counter()
    m1, s1 = fid_oracle.calculate_activation_statistics(x)
    t_cov = time.perf_counter() - t0
    m2, s2 = fid_oracle.calculate_activation_statistics(_cases.pool3_like_features(3000, dims, 43, shift=0.1))
    t0 = time.perf_counter()
    fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    t_fd = time.perf_counter() - t0
    lg = np.random.default_rng(0).standard_normal((n_job, 1000)).astype(np.float32)
    t_is, _ = _median_time(lambda: is_oracle.inception_score_from_logits(lg, is_oracle.T_COCO, 10, "coco", dtype=np.float32),
                           repeats=3, warmup=1)
    torch.set_num_threads(all_threads)
    n_used = (n // bs) * bs
    per_img_dec, per_img_fwd = t_dec / n_used, t_fwd / n
    # the reference overlaps the loader workers with the forward pass: per image the slower of the two stages
    job_s = n_job * max(per_img_dec, per_img_fwd) + t_cov + t_fd + t_is
    base = {
        "value": n_job / job_s, "unit": "images/sec", "cores": threads, "kind": "port",
        "sample": (f"oracle/ on {n} images of the timed set, batch {bs}, 1 warm-up + median of {r_fwd} repeats: PNG decode + PIL "
                   f"resize + ToTensor on {workers} DataLoader workers {per_img_dec * 1e3:.2f} ms/img, InceptionV3 fp32 forward "
                   f"(torch CPU, {threads} threads) {per_img_fwd * 1e3:.2f} ms/img; at the job's size ({n_job} x {dims}): "
                   f"np.mean/np.cov {t_cov:.2f} s, reference-form Frechet distance (scipy sqrtm) {t_fd:.2f} s, IS* reduction "
                   f"{t_is:.3f} s; job time = N x max(decode, forward) + cov + sqrtm + IS (loader overlapped as in the reference)"),
        "stages": {"decode_resize_ms_per_img": per_img_dec * 1e3, "forward_ms_per_img": per_img_fwd * 1e3,
                   "cov_s": t_cov, "frechet_s": t_fd, "is_reduce_s": t_is, "repeats": {"decode": r_dec, "forward": r_fwd}},
        "host_cpus": os.cpu_count(), "host_usable_cpus": usable_cpus(), "loader_workers": workers,
    }
    return base, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=60)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--images", type=int, default=JOB_IMAGES, help="images of the job (the metric's 30 000)")
    ap.add_argument("--batch", type=int, default=0,
                    help="explicit images per step: the job becomes steps x batch images (0: the job is --images, a step = images / steps)")
    ap.add_argument("--device-batch", type=int, default=0, help="images per device batch (0: chosen by rank_batch)")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong")
    ap.add_argument("--ref-images", type=int, default=JOB_IMAGES,
                    help="images of the reference side (SURVEY 8(d) Config 3: 30k vs 30k); sharded over the ranks and all-reduced "
                         "like the generated side, before the timed region (the README recipe passes this side as an .npz)")
    ap.add_argument("--collective", choices=["auto", "off"], default="auto",
                    help="N = 1: auto brings up a ONE-rank RCCL group (backend nccl) so that the job's all-reduces really execute "
                         "on the single GPU (config.collective records backend and times); off: no process group at N = 1")
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip cpu_baseline AND parity (both need the CPU oracle)")
    ap.add_argument("--cpu-sample", type=int, default=250, help="images of the timed set the CPU oracle processes")
    ap.add_argument("--no-cross-check", action="store_true")
    ap.add_argument("--no-kernel-probe", action="store_true", help="skip the stand-alone timing of the HBM-bound kernels after the timed region")
    ap.add_argument("--channels-last", type=int, default=-1)
    ap.add_argument("--no-host-feed", action="store_true",
                    help="skip the host_feed leg (N = 1, after the timed region: the same job fed from pinned HOST memory in "
                         "--feed-batch slices, as the CLI's --u8-cache path feeds it; reported next to `value`, never instead of it)")
    ap.add_argument("--feed-batch", type=int, default=50, help="loader batch of the host_feed leg (README.md:214-219: 50)")
    ap.add_argument("--no-cli-process", action="store_true",
                    help="skip the cli_process object (N = 1, after the timed region: the README recipe `python -m tise_toolbox_amd.fid_score "
                         "--path1 ref.npz --path2 DIR` run as a FRESH CHILD PROCESS on the png_feed files, wall clock + TISE_TIMING phases)")
    ap.add_argument("--png-images", type=int, default=30000,
                    help="png_feed leg (N = 1, after the timed region, never `value`): this many images of the timed set are written as "
                         "PNG files and the job is run from the FILES through the CLIs' feed (png_ring.py); 0 skips the leg")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(_self_launch(args.gpus))                  # before anything touches the GPU in this process
    # The contract is ONE JSON line on stdout.  librccl prints a version banner ("RCCL version : ...", five lines) to the C
    # stdout when its first communicator comes up (seen on the GPU boxes, flushed at exit): file descriptor 1 is pointed at
    # stderr for the life of the process and the JSON line is written to a duplicate of the original descriptor.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    from tise_toolbox_amd import _lib, device, dist as tdist, fid_score
    from tise_toolbox_amd.engine import RealismEngine, T_COCO, frechet_solver
    coll_note = None
    t_init0 = time.perf_counter()
    if "WORLD_SIZE" not in os.environ and args.gpus == 1 and args.collective == "auto" and torch.cuda.device_count() > 0:
        # one GPU: the collective leg of the job still executes -- a one-rank "nccl" (= RCCL) group through the product's
        # own init path; a box whose RCCL cannot come up falls back to no group and says so in config.collective
        try:
            rank, world, local_rank = tdist.init_from_env(force=True)
        except Exception as e:                                           # noqa: BLE001
            coll_note = f"one-rank RCCL group failed to initialise: {type(e).__name__}: {e}"[:300]
            rank, world, local_rank = 0, 1, 0
    else:
        rank, world, local_rank = tdist.init_from_env()
    t_group_init = time.perf_counter() - t_init0
    if world != args.gpus:
        raise SystemExit(f"[bench] --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    backend = torch.distributed.get_backend() if torch.distributed.is_initialized() else None
    if os.environ.get("TISE_FORCE_DEVICE0"):       # testing aid: several ranks on one GPU (gloo backend only)
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cl = None if args.channels_last < 0 else bool(args.channels_last)
    eng = RealismEngine(dims=2048, device_index=local_rank, seed=0, with_logits=True, channels_last=cl)
    K, W = args.steps, args.warmup
    n_job = K * args.batch if args.batch > 0 else args.images
    B = n_job / K                                             # images per step (a step = 1/K of the job)
    if args.scaling == "strong":
        n_total = n_job
        lo, hi = tdist.shard_range(n_total, rank, world)
    else:
        n_total = n_job * world
        lo, hi = rank * n_job, (rank + 1) * n_job
    n_rank = hi - lo
    rb = args.device_batch if args.device_batch > 0 else rank_batch(n_rank)
    chunks = [(a, min(a + rb, n_rank)) for a in range(0, n_rank, rb)]

    # ---- inputs resident in HBM before the timed region ------------------------------------------
    data = torch.empty((n_rank, 256, 256, 3), dtype=torch.uint8, device=dev)
    for i in range(0, n_rank, 1000):
        j = min(i + 1000, n_rank)
        data[i:j] = synth_images_device(lo + i, lo + j, dev, seed=0)
    # reference side (the README recipe passes it as an .npz, fid_score.py:200-203): before the timed region, but computed as
    # SURVEY 8(d) Config 3 says -- --ref-images (30 000) images sharded over the ranks exactly like the generated side, ONE
    # all-reduce of {n, s, S} for this side (the generated side's is inside the timed region)
    r_lo, r_hi = tdist.shard_range(args.ref_images, rank, world)
    eng.begin(n_total=args.ref_images)
    t_ref0 = time.perf_counter()
    for i in range(r_lo, r_hi, 1000):
        j = min(i + 1000, r_hi)
        eng.step_u8(synth_images_device(i, j, dev, seed=1, shift=0.12), i)
    torch.cuda.synchronize()
    t_ref1 = time.perf_counter()
    eng.reduce()                                    # first collective of the process: also creates the RCCL communicator
    torch.cuda.synchronize()
    t_ref2 = time.perf_counter()
    mu_ref, sigma_ref = eng.statistics()
    ref_side = {"images": args.ref_images, "images_this_rank": r_hi - r_lo, "seconds_images": t_ref1 - t_ref0,
                "allreduce_incl_communicator_creation_ms": (t_ref2 - t_ref1) * 1e3,
                "note": "sharded over the ranks and all-reduced like the generated side; outside the timed region"}
    solver = frechet_solver(2048, dev)
    solver.set_profiling(True)

    # ---- warmup (allocator, clocks, RCCL communicator) ----------------------------------------------
    eng.begin(n_total=max(W, 1) * rb)
    for s in range(W):
        a, b = chunks[s % len(chunks)]
        eng.step_u8(data[a:b], s * rb)
    if W > 0:
        eng.reduce()                                # also brings up the RCCL communicator outside the timed region
        mu_w, sig_w = eng.statistics()
        solver.distance(mu_w, sig_w, mu_ref, sigma_ref)
        eng.inception_score()
    elif world > 1:
        tdist.all_reduce_sum_(torch.zeros(1024, dtype=torch.float64, device=dev))
    torch.cuda.synchronize()

    # ---- timed region -------------------------------------------------------------------------------
    nch = len(chunks)
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(nch)]
    ev_tail = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    from tise_toolbox_amd.conv_split import SplitConv as _SC
    # HIP events around every convolution launch cost ~1.5 % of a step (130 extra records per 65 launches), so they
    # bracket the conv launches of at most three device batches -- first, middle, last: 195 launches, plenty for the
    # average -- whatever the number of steps (measured: 0.8 % of the job with five of ten batches bracketed); the
    # rocprofv3 summary of the same command is the cross-check.
    # TISE_BENCH_MODE=noevents switches them off.
    mode = os.environ.get("TISE_BENCH_MODE", "events")
    conv_timer = [] if mode == "events" else None
    evented = {0, (nch - 1) // 2, nch - 1}
    _SC.timer = None
    u8_stem = getattr(eng, "_u8_stem", False)
    eng.begin(n_total=n_total, temperature=T_COCO, splits=10, rule="coco")
    # after the second device batch when there are at least three (the host blocks until the side stream has the factor:
    # two batches are queued on the main stream by then); with one or two batches per rank (8 GPUs: 2 x 1875 images) after
    # the first, so that the factorisation still runs under a batch instead of in the tail
    prefactor_after = (1 if nch >= 3 else 0) if os.environ.get("TISE_BENCH_PREFACTOR", "1") != "0" else -1
    tdist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s, (a, b) in enumerate(chunks):
        batch = data[a:b]
        _SC.timer = conv_timer if (conv_timer is not None and s in evented) else None
        ev[s][0].record()
        if u8_stem:
            x = device.resize_u8_only(batch, (299, 299))       # uint8 out; the stem conv applies the input table
            ev[s][1].record()
            feats, logits = eng._trunk_u8(x)
        else:
            x = device.resize_bilinear_u8(batch, (299, 299), eng.lut, channels_last=eng.channels_last)
            ev[s][1].record()
            feats, logits = eng._trunk(x, prenormalized=True)
        ev[s][2].record()
        eng.stats.update(feats)                              # ONE launch: fp64-MFMA S += X^T X, column sums, row count
        ev[s][3].record()
        eng.is_acc.update(logits, lo + a)
        if s == prefactor_after:
            # the reference statistics are an input: their pivoted Cholesky does not depend on the generated set, so it
            # runs on a side stream under the network passes (the host waits for that stream only; two device batches
            # are already queued on the main stream)
            solver.prefactor(sigma_ref)
    _SC.timer = None
    ev_tail[0].record()
    t_loop_host = time.perf_counter()
    eng.reduce()                                             # RCCL all-reduce of {n, s, S} and the IS* sums
    if backend is not None:
        torch.cuda.current_stream().synchronize()            # so that allreduce_ms is the collective, not the queue
    ev_tail[1].record()
    t_reduce_host = time.perf_counter()
    mu, sigma = eng.statistics()
    ev_tail[2].record()
    if prefactor_after >= 0:
        res = solver.distance_prefactored(mu_ref, mu, sigma)  # GEMM + tridiagonalisation + bisection; one 8-double read
        if res["flags"] & _lib.TISE_FLAG_NONFINITE:           # fid_score.py:156-160 rescue: full path with eps
            res = solver.distance(mu, sigma, mu_ref, sigma_ref, 1e-6)
    else:
        res = solver.distance(mu, sigma, mu_ref, sigma_ref)
    is_mean, is_std = eng.inception_score()
    ev_tail[3].record()
    tdist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()

    elapsed = torch.tensor([t1 - t0], dtype=torch.float64, device=dev)
    if world > 1:
        torch.distributed.all_reduce(elapsed, op=torch.distributed.ReduceOp.MAX)
    elapsed = float(elapsed.item())
    # per-rank breakdown: with only the max-over-ranks time a straggler or a slow communicator cannot be
    # told from a slow trunk.  loop_device = first launch of the image loop -> last (HIP events on this rank's stream);
    # reduce = host wall from the end of the loop's enqueue to the completed all-reduce (it contains the wait for the device
    # to finish the loop); first_collective = the reference side's all-reduce, communicator creation included
    mine = torch.tensor([float(rank), n_rank, ev[0][0].elapsed_time(ev_tail[0]) * 1e-3, t_loop_host - t0, t_reduce_host - t_loop_host,
                         ev_tail[0].elapsed_time(ev_tail[1]) * 1e-3, t_ref2 - t_ref1, t1 - t0, t_group_init], dtype=torch.float64, device=dev)
    per_rank = [mine]
    if world > 1:
        if backend != "nccl":                                            # gloo (tests: several ranks on one GPU) gathers host tensors
            mine = mine.cpu()
        per_rank = [torch.empty_like(mine) for _ in range(world)]
        torch.distributed.all_gather(per_rank, mine)
    rank_rows = [dict(zip(("rank", "images", "loop_device_s", "loop_host_enqueue_s", "reduce_host_s", "reduce_device_s",
                           "first_collective_s", "total_s", "init_process_group_s"), [float(v) for v in t.tolist()])) for t in per_rank]
    for r_ in rank_rows:
        r_["rank"], r_["images"] = int(r_["rank"]), int(r_["images"])

    from tise_toolbox_amd.trunk import SplitTrunk
    conv_events = conv_timer or []
    timed_steps = len(evented) if conv_timer is not None else 0
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
        full = [s for s, (a, b) in enumerate(chunks) if b - a == rb]
        resize_ms = float(np.mean([ev[s][0].elapsed_time(ev[s][1]) for s in full]))
        trunk_ms = float(np.mean([ev[s][1].elapsed_time(ev[s][2]) for s in full]))
        syrk_ms = float(np.mean([ev[s][2].elapsed_time(ev[s][3]) for s in full]))
        phases = solver.phase_ms()
        d = 2048
        tiles = d // 64
        syrk_flop = 2.0 * rb * 64 * 64 * (tiles * (tiles + 1) // 2)        # upper 64x64 tiles only, per launch
        resize_bytes = rb * (256 * 256 * 3 + 299 * 299 * 3 * (1 if u8_stem else 4))
        kern = {
            "syrk_f32_upper_bk64_kernel<colsum>": {"bound": "mfma", "achieved": syrk_flop / (syrk_ms * 1e-3) / 1e12,
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
                "instances": "all convolution launches of a device batch, as rocprofv3 lists them (conv_split_fast_kernel<TN>, "
                             "conv_split_rowwin_kernel<TN, NP, POOLH>, conv_regw32_kernel<COUT, .., POOL>, conv_poolin_kernel<TNW, VT>; "
                             "profiles/r04*_kernel_stats.md)",
                "clock_note": "the chip is power-limited on this loop: the bare inner loop (LDS fragment reads + 3 MFMAs per product, "
                              "nothing else; tools/probes/mfma_shape_probe.hip, profiles/r03k_mfma_shape_probe.txt) sustains "
                              "1.57-1.69 PFLOP/s of fp16 MFMA with v_mfma_f32_16x16x32_f16 at 1.79 GHz (1.39-1.48 at 1.52 GHz with "
                              "32x32x16) of the 2.5 PFLOP/s datasheet peak at 2.4 GHz; peak below is the datasheet figure / 3",
                "sustained_mfma_f16_probe": SUSTAINED_F16_MFMA_TFLOPS,
                "frac_of_sustained": 3.0 * conv_flop / (conv_ms * 1e-3) / 1e12 / SUSTAINED_F16_MFMA_TFLOPS,
                "bound": "mfma", "achieved": conv_flop / (conv_ms * 1e-3) / 1e12, "peak": PEAK_F16_MFMA_TFLOPS / 3.0,
                "unit": "TFLOP/s", "avg_ms": conv_ms / timed_steps, "avg_launch_ms": conv_ms / n_launch,
                "launches_per_batch": n_launch / timed_steps, "algorithmic_flop_per_batch": conv_flop / timed_steps,
                "batches_with_events": timed_steps,
                "mfma_tflops_f16": 3.0 * conv_flop / (conv_ms * 1e-3) / 1e12, "mfma_peak_f16": PEAK_F16_MFMA_TFLOPS}
        if not args.no_kernel_probe:
            kern.update(hbm_kernel_probe(eng, data[chunks[0][0]:chunks[0][1]], dev))
        for k in kern.values():
            k["frac"] = k["achieved"] / k["peak"]
        traffic, traffic_src = None, None
        pmc_path = os.path.join(ROOT, "profiles", "pmc_summary.json")
        dom = max((k for k in kern if not kern[k].get("probe")), key=lambda k: kern[k]["avg_ms"])
        if os.path.exists(pmc_path):
            try:
                pj = json.load(open(pmc_path))
                traffic = pj.get(dom, {}).get("hbm_bytes_per_launch")
                traffic_src = pj.get("_source")
            except Exception:
                traffic = None
        roofline = {"kernel": dom, "bound": kern[dom]["bound"], "achieved": kern[dom]["achieved"],
                    "peak": kern[dom]["peak"], "unit": kern[dom]["unit"], "frac": kern[dom]["frac"],
                    "traffic": traffic, "traffic_source": traffic_src,
                    "avg_launch_ms": kern[dom].get("avg_launch_ms", kern[dom]["avg_ms"]), "kernels": kern}
        allreduce_ms = ev_tail[0].elapsed_time(ev_tail[1])
        out = {
            "metric": "images/sec through InceptionV3+FID on 30k 256x256 @1/2/4/8 GPU; |dFID| vs ref",
            "value": n_total / elapsed, "unit": "images/sec", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": elapsed / K * 1e3, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
            "dtype": conv_dtype, "data": "synthetic",
            "config": {"workload": f"IS*+FID of ONE set of {n_total} synthetic 256x256 images (BASELINE configs[1]/[2]: 30k "
                                   f"images, InceptionV3 pool3 2048-d) sharded over {world} GPU(s): {n_rank} images per GPU in "
                                   f"device batches of {rb}; inputs are pre-decoded uint8 pixels resident in HBM (PNG decode and "
                                   f"host->device copy excluded from the timed region); seeded stand-in InceptionV3 weights, "
                                   f"reference stats from {args.ref_images} images",
                       "step_images": B, "images_per_gpu": n_rank, "images_total": n_total, "device_batch": rb,
                       "device_batches_per_gpu": nch, "dims": 2048, "trunk": trunk_desc, "parallelism": f"dp{world}",
                       "collective": {"world_size": world, "backend": backend,
                                      "forced_one_rank_group": bool(world == 1 and backend is not None),
                                      "init_process_group_s": t_group_init if backend is not None else None,
                                      "all_reduces": "one per side: [S | s | n] (33.57 MB fp64) + the IS* sums (80 KB) of the generated "
                                                     "side inside the timed region (allreduce_ms); the reference side's before it "
                                                     "(reference_side)", "note": coll_note,
                                      "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
                                      "launcher": os.environ.get("TISE_BENCH_LAUNCHER", "external" if world > 1 else "none")}},
            "reference_side": ref_side,
            "ranks": rank_rows,
            "roofline": roofline,
            "stage_ms_per_device_batch": {"resize": resize_ms, "trunk": trunk_ms, "cov_syrk": syrk_ms},
            "allreduce_ms": allreduce_ms,
            "finalize_ms": {"host_wall_after_loop": (t1 - t_loop_host) * 1e3,
                            "host_wall_allreduce": (t_reduce_host - t_loop_host) * 1e3,
                            "stats_finalize": ev_tail[1].elapsed_time(ev_tail[2]),
                            "frechet_plus_is": ev_tail[2].elapsed_time(ev_tail[3]), **{k: v for k, v in phases.items()},
                            "pchol_side_stream_ms": (solver.prefactor_ms() if prefactor_after >= 0 else None)},
            "scores": {"fid": float(res["fid"]), "is_mean": is_mean, "is_std": is_std, "rank": res["rank"],
                       "flags": res["flags"]},
            "trunk_tflops": 11.42e9 * rb / (trunk_ms * 1e-3) / 1e12,
        }
        out["parity"] = None
        out["cpu_baseline"] = None
        out["cross_check"] = None
        out["host_feed"] = None
        if world == 1 and not args.no_host_feed:
            out["host_feed"] = host_feed_leg(eng, data, lo, n_total, args.feed_batch, mu_ref, sigma_ref, solver, dev,
                                             float(res["fid"]), n_total / elapsed)
        out["png_feed"] = None
        out["cli_process"] = None
        if world == 1 and args.png_images > 0:
            # the extra legs must never cost the line its headline: a failure is recorded, not raised
            try:
                out["png_feed"] = png_feed_leg(eng, data, lo, min(args.png_images, n_rank), args.feed_batch, mu_ref, sigma_ref, solver, dev,
                                               cli=not args.no_cli_process)
                out["cli_process"] = out["png_feed"].pop("cli_process", None)
                # the in-leg resident reference is ONE run of the same 30 000 images and moves by a few per cent between runs;
                # the line's headline rate is K timed steps: the steadier denominator, reported beside it
                out["png_feed"]["ratio_to_value"] = out["png_feed"]["images_per_s"] / (n_total / elapsed)
            except Exception as e:                                       # noqa: BLE001
                import traceback
                traceback.print_exc()
                out["png_feed"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        if world == 1 and not args.no_cpu_baseline:
            # ---- CPU oracle on the first images of the timed set: the baseline AND the parity figures --------
            n_cpu = max(50, (min(args.cpu_sample, n_rank) // 50) * 50)
            gen_u8 = data[:n_cpu].cpu().numpy()
            ref_dev = synth_images_device(0, n_cpu, dev, seed=1, shift=0.12)
            base, feats_cpu = cpu_reference_pass(gen_u8, ref_dev.cpu().numpy(), n_total)
            out["cpu_baseline"] = base
            from oracle import fid_oracle, is_oracle

            def dev_pass(u8):
                fs, ls = [], []
                for i in range(0, n_cpu, 50):
                    f, l = eng.features_from_u8(u8[i:i + 50])
                    fs.append(f); ls.append(l)
                return torch.cat(fs), torch.cat(ls)
            fg, lg = dev_pass(data[:n_cpu])
            fr, _ = dev_pass(ref_dev)
            sg, sr = device.StatsAccumulator(2048, dev), device.StatsAccumulator(2048, dev)
            sg.update(fg); sr.update(fr)
            fid_dev = float(fid_score.calculate_frechet_distance(*sg.finalize(), *sr.finalize()))
            is_dev = fid_score_is(lg, dev)
            (fg_c, lg_c), (fr_c, _) = feats_cpu["gen"], feats_cpu["ref"]
            fid_cpu = float(fid_oracle.calculate_frechet_distance(*fid_oracle.calculate_activation_statistics(fg_c),
                                                                  *fid_oracle.calculate_activation_statistics(fr_c)))
            is_cpu = is_oracle.inception_score_from_logits(lg_c, is_oracle.T_COCO, 10, "coco", dtype=np.float32)
            out["parity"] = {
                "n": n_cpu, "against": "oracle/ (CPU: PIL-exact resize, fp32 InceptionV3, np.cov, scipy sqrtm; fp32 IS* reduction) on "
                                       "the first n images of the timed set vs the first n reference images, same weights",
                "fid_device": fid_dev, "fid_oracle": fid_cpu, "dfid": abs(fid_dev - fid_cpu),
                "is_device": is_dev[0], "is_oracle": is_cpu[0], "dis": abs(is_dev[0] - is_cpu[0]),
                "dis_std": abs(is_dev[1] - is_cpu[1]),
                "max_feature_err_rel": float(np.abs(fg.cpu().numpy() - fg_c).max() / np.abs(fg_c).max()),
                "tolerance": {"dfid": 1e-3, "dis": 1e-4}}
        if world == 1 and not args.no_cross_check and isinstance(eng.fused, SplitTrunk):
            cc_chunks = [(a, min(a + 1000, n_rank)) for a in range(0, n_rank, 1000)]      # MIOpen's fp32 activations: 1000 images at a time
            out["cross_check"] = cross_check_fp32(eng, data, cc_chunks, lo, n_total, mu, sigma, mu_ref, sigma_ref,
                                                  float(res["fid"]), (is_mean, is_std), solver, dev)
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    tdist.barrier()
    tdist.shutdown()


def host_feed_leg(eng, data, lo, n_total, feed_batch, mu_ref, sigma_ref, solver, dev, fid_resident, rate_resident):
    """The same job with the images in page-locked HOST memory, inside the timed region: the loader hands over
    ``feed_batch`` images at a time (the README recipe's --batch-size 50), engine.coalesce_u8 gathers them into device
    batches on a side stream (the mechanism behind the CLIs' loaders since round 4: --batch-size only defines the
    drop-last rule), the trunk runs once per device batch.  Whole job = image loop + reduce + finalize + Frechet + IS*,
    exactly the timed region of `value`.  PNG decode is not part of it (profiles/r04*_cli_host_inclusive.txt)."""
    from tise_toolbox_amd.engine import T_COCO, coalesce_u8, device_batch_images
    n = data.shape[0]
    host = torch.empty(tuple(data.shape), dtype=torch.uint8).pin_memory()
    for i in range(0, n, 2000):
        host[i:i + 2000].copy_(data[i:i + 2000])
    torch.cuda.synchronize()
    limit = device_batch_images(feed_batch, int(np.prod(data.shape[1:])))
    n_used = (n // feed_batch) * feed_batch                 # drop-last rule of the loader batch (n divides: 30 000 / 50)

    def run():
        eng.begin(n_total=n_total, temperature=T_COCO, splits=10, rule="coco")
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        base = lo
        for big in coalesce_u8((host[i:i + feed_batch] for i in range(0, n_used, feed_batch)), dev, limit):
            eng.step_u8(big, base)
            base += big.shape[0]
        t_loop = time.perf_counter()
        eng.reduce()
        mu, sigma = eng.statistics()
        r = solver.distance(mu, sigma, mu_ref, sigma_ref)
        eng.inception_score()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        return t1 - t0, t_loop - t0, float(r["fid"])
    run()                                                   # staging buffers, allocator
    wall, host_loop, fid = run()
    return {"images_per_s": n_used / wall, "seconds": wall, "host_loop_seconds": host_loop, "feed_batch": feed_batch,
            "device_batch": limit, "images": n_used, "pinned_host_bytes": int(host.numel()),
            "ratio_to_resident": (n_used / wall) / rate_resident, "fid": fid, "dfid_vs_resident": abs(fid - fid_resident),
            "note": "whole job from pinned host memory (H2D copies inside the timed region, gathered into device batches on a side "
                    "stream); the Frechet solve here is the one-call form (no side-stream prefactor), `value` stays the device-resident rate"}


def _write_pngs(args):
    npy, lo, hi, d = args
    from PIL import Image
    arr = np.load(npy, mmap_mode="r")
    for i in range(lo, hi):
        Image.fromarray(np.asarray(arr[i])).save(os.path.join(d, f"{i:06d}.png"))
    return hi - lo


def cli_process_leg(png_dir, n, feed_batch, mu_ref, sigma_ref, tmp, fid_feed):
    """What a USER waits for: the README recipe (README.md:214-219 of the reference: fid_score.py
    --batch-size 50 --path1 <stats .npz> --path2 <image dir>) as a fresh child process -- interpreter start, imports, HIP
    context, model + engine, the image loop from PNG files, the Frechet distance, exit -- timed from outside with a wall
    clock, twice (the second run finds the page cache, the stand-in cache and the code objects warm).  The child is started
    with subprocess (a new process; this one is never replaced).  Never `value`."""
    import subprocess
    ref = os.path.join(tmp, "ref_stats.npz")
    np.savez(ref, mu=mu_ref.cpu().numpy(), sigma=sigma_ref.cpu().numpy())
    outf = os.path.join(tmp, "cli_out.txt")
    cmd = [sys.executable, "-m", "tise_toolbox_amd.fid_score", "--batch-size", str(feed_batch), "--path1", ref, "--path2", png_dir,
           "--synthetic-weights", "--saved_file", outf]
    env = dict(os.environ, TISE_TIMING="1")
    env["PYTHONPATH"] = os.pathsep.join([ROOT] + [p for p in env.get("PYTHONPATH", "").split(os.pathsep) if p])
    runs = []
    for _ in range(2):
        t0 = time.perf_counter()
        r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
        wall = time.perf_counter() - t0
        phases = {}
        for ln in r.stderr.splitlines():
            if ln.startswith("[tise timing] ") and " s after process start" in ln:
                label, rest = ln[len("[tise timing] "):].rsplit(": ", 1)
                phases[label.split(" (")[0]] = float(rest.split(" s after")[0])
        feed = [ln for ln in r.stderr.splitlines() if ln.startswith("[tise] png feed")]
        fid = None
        try:
            fid = float(open(outf).read().split("FID:")[1].split()[0])
        except Exception:                                                  # noqa: BLE001
            pass
        runs.append({"seconds": wall, "returncode": r.returncode, "phases_s_after_process_start": phases,
                     "feed_line": feed[-1][:400] if feed else None, "fid": fid,
                     "stderr_tail": None if r.returncode == 0 else r.stderr[-600:]})
    ok = all(x["returncode"] == 0 for x in runs)
    return {"command": "python -m tise_toolbox_amd.fid_score --batch-size %d --path1 ref_stats.npz --path2 <dir of %d PNG files> --synthetic-weights"
                       % (feed_batch, n),
            "images": n, "seconds": runs[0]["seconds"], "seconds_second_run": runs[1]["seconds"],
            "images_per_s": n / runs[0]["seconds"] if ok else None, "runs": runs,
            "dfid_vs_png_feed": (abs(runs[1]["fid"] - fid_feed) if ok and runs[1]["fid"] is not None else None),
            "note": "whole fresh child process, wall clock around subprocess.run: interpreter + imports + HIP context + model / engine + image "
                    "loop from PNG files + Frechet distance + exit; phases from TISE_TIMING=1 (seconds after process start).  The second run starts "
                    "right after the first one freed its device memory (~16 GiB with the feeds' 1000-image device batches; ~45 GiB with round 5's 3000): when the driver "
                    "is still clearing that memory, the new process's first big allocation waits for it (~45 GB/s: up to ~0.3 s now, ~1 s then; DESIGN.md section 4f) -- `seconds` is the common case, `seconds_second_run` the "
                    "back-to-back case"}


def png_feed_leg(eng, data, lo, n, feed_batch, mu_ref, sigma_ref, solver, dev, cli=True):
    """Row a2 inside the bench: the first ``n`` images of the timed set are written as PNG FILES (Pillow, its default
    compression) and the job -- image loop + reduce + finalize + Frechet + IS* -- is run from the files through the feed of
    the drop-in CLIs: tise_toolbox_amd.png_ring (decode processes, csrc/png_decode.c -> shared page-locked ring ->
    side-stream H2D -> device batches).  PNG decode, the H2D copies and the start of the decode processes are INSIDE the
    timed wall; writing the files is not.  Never `value`."""
    import shutil
    from concurrent.futures import ProcessPoolExecutor
    from tise_toolbox_amd import png_ring
    from tise_toolbox_amd.engine import T_COCO, device_batch_images
    n = (n // feed_batch) * feed_batch
    tmp = tempfile.mkdtemp(prefix="tise_bench_pngfeed_")
    try:
        npy = os.path.join(tmp, "pixels.npy")
        np.save(npy, data[:n].cpu().numpy())
        d = os.path.join(tmp, "png")
        os.makedirs(d)
        procs = max(1, png_ring.usable_cpus())
        step = -(-n // (4 * procs))
        t0 = time.perf_counter()
        with ProcessPoolExecutor(procs) as ex:
            list(ex.map(_write_pngs, [(npy, a, min(a + step, n), d) for a in range(0, n, step)]))
        t_write = time.perf_counter() - t0
        files = [os.path.join(d, f"{i:06d}.png") for i in range(n)]
        png_bytes = sum(os.path.getsize(f) for f in files[:200]) / 200.0
        limit = device_batch_images(feed_batch, int(np.prod(data.shape[1:])))

        def resident():
            eng.begin(n_total=n, temperature=T_COCO, splits=10, rule="coco")
            for a in range(0, n, limit):
                eng.step_u8(data[a:min(a + limit, n)], lo + a)
            eng.reduce()
            mu, sigma = eng.statistics()
            r = solver.distance(mu, sigma, mu_ref, sigma_ref)
            eng.inception_score()
            torch.cuda.synchronize()
            return float(r["fid"])

        def from_files():
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            loader = png_ring.PngRingLoader(files, feed_batch, dev, group=limit // feed_batch)     # starts the decode processes
            eng.begin(n_total=n, temperature=T_COCO, splits=10, rule="coco")
            base = lo
            for big in loader:
                eng.step_u8(big, base)
                base += big.shape[0]
            t_loop = time.perf_counter()
            eng.reduce()
            mu, sigma = eng.statistics()
            r = solver.distance(mu, sigma, mu_ref, sigma_ref)
            eng.inception_score()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            return t1 - t0, t_loop - t0, float(r["fid"]), loader
        fid_res = resident()
        t0 = time.perf_counter()
        resident()
        t_res = time.perf_counter() - t0
        from_files()                                                # page cache, pinned-ring registration path, allocator
        wall, loop, fid, loader = from_files()
        cli_obj = None
        if cli:
            try:
                cli_obj = cli_process_leg(d, n, feed_batch, mu_ref, sigma_ref, tmp, fid)
            except Exception as e:                                       # noqa: BLE001 -- a child that hangs or dies is a recorded failure of this leg only
                cli_obj = {"error": f"{type(e).__name__}: {e}"[:400]}
        return {"cli_process": cli_obj, "images_per_s": n / wall, "seconds": wall, "image_loop_seconds": loop, "images": n, "feed_batch": feed_batch,
                "device_batch": limit, "device_batches": loader.item_sizes() if hasattr(loader, "h") else None,
                "decode_processes": loader.workers, "native_workers": loader.native is not None,
                "row_filters": "device (csrc/png_unfilter.hip; the workers only inflate)" if loader.framed else "host",
                "host_hardware_threads": os.cpu_count(),
                "host_usable_cpus": png_ring.usable_cpus(), "all_decoded_after_s": loader.decode_seconds,
                "feeder_waited_s": {"for_decode": loader.wait_decode_seconds, "for_a_device_buffer": loader.wait_buffer_seconds,
                                    "in_memcpy_calls": loader.enqueue_seconds, "for_copies_to_land": loader.wait_copy_seconds},
                "png_bytes_per_image": png_bytes, "write_seconds_untimed": t_write,
                "resident_same_images_per_s": n / t_res, "ratio_to_resident": (n / wall) / (n / t_res),
                "fid": fid, "dfid_vs_resident": abs(fid - fid_res),
                "note": "whole job from PNG FILES: start of the decode processes (csrc/png_worker.c) + zlib inflate (csrc/png_decode.c, Pillow "
                        "for files outside its subset) + H2D + PNG row filters on the device (csrc/png_unfilter.hip) + image loop + reduce + finalize + Frechet (one-call form) + IS*; the same images from HBM, same "
                        "code path, alongside (resident_same_images_per_s); host_usable_cpus = affinity and cgroup CPU quota"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def hbm_kernel_probe(eng, batch_u8, dev, reps=20):
    """The HBM-bound hand-written kernels of a device batch, each launched ALONE `reps` times back to back between two
    HIP events AFTER the timed region (the convolution and covariance kernels are timed inside it): the remaining
    max-pools (Mixed_6a / 7a pool branches; stem pool 1 is taken in Conv2d_2b's epilogue, pool 2 in Conv2d_4a's epilogue and
    conv_poolin_kernel's operand load), the average-pool tails,
    the global mean, the IS* row / column kernels, stats finalize, and the resize.  `achieved` = ALGORITHMIC bytes
    (one read of the input + one write of the output) / average launch time."""
    from tise_toolbox_amd import device
    from tise_toolbox_amd.trunk import SplitTrunk
    n = batch_u8.shape[0]
    out = {}

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    def add(name, nbytes, ms, note):
        out[name] = {"bound": "hbm", "achieved": nbytes / (ms * 1e-3) / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "avg_ms": ms,
                     "algorithmic_bytes_per_launch": nbytes, "probe": f"alone, {reps} back-to-back launches after the timed region; {note}"}

    if isinstance(eng.fused, SplitTrunk):
        for (h, w, c, co, name) in ((35, 35, 288, 768, "maxpool3s2_split_kernel[35x35x288]"), (17, 17, 768, 1280, "maxpool3s2_split_kernel[17x17x768]")):
            x = torch.rand((n, h, w, 2 * c), device=dev).half()
            oh, ow = (h - 3) // 2 + 1, (w - 3) // 2 + 1
            o = torch.empty((n, oh, ow, 2 * co), dtype=torch.float16, device=dev)
            ms = timed(lambda: SplitTrunk._maxpool_split(x, o, co - c))
            add(name, n * (h * w + oh * ow) * c * 4, ms, f"split tensor {n}x{h}x{w}x{c} -> slice of {co} channels")
        for (h, w, c, co, name) in ((35, 35, 64, 288, "avgpool3_bias_relu_split_kernel[35x35x64]"), (17, 17, 192, 768, "avgpool3_bias_relu_split_kernel[17x17x192]"),
                                    (8, 8, 192, 2048, "avgpool3_bias_relu_split_kernel[8x8x192]")):
            raw = torch.rand((n, h, w, c), device=dev)
            bias = torch.rand(c, device=dev)
            o = torch.empty((n, h, w, 2 * co), dtype=torch.float16, device=dev)
            ms = timed(lambda: SplitTrunk._avgpool_split(raw, bias, o, co - c))
            add(name, 2 * n * h * w * c * 4, ms, f"raw fp32 {n}x{h}x{w}x{c} -> split slice")
        a = torch.rand((n, 8, 8, 2 * 2048), device=dev).half()
        feat = torch.empty((n, 2048), dtype=torch.float32, device=dev)
        from tise_toolbox_amd import _lib
        import ctypes
        st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        ms = timed(lambda: _lib.call("tise_split_mean_nhwc", ctypes.c_void_p(a.data_ptr()), n, 64, 2048, ctypes.c_void_p(feat.data_ptr()), st()))
        add("split_mean_kernel", n * (64 * 2048 * 4 + 2048 * 4), ms, f"split tensor {n}x8x8x2048 -> fp32 {n}x2048")
    ms = timed(lambda: device.resize_u8_only(batch_u8, (299, 299)))
    add("resize_bilinear_u8_kernel[probe]", n * (256 * 256 * 3 + 299 * 299 * 3), ms, "uint8 256x256x3 -> uint8 299x299x3")
    logits = torch.randn((n, 1000), device=dev)
    acc = device.InceptionScoreAccumulator(1000, n, 0.9091363549232483, 10, "coco", False, dev)
    ms = timed(lambda: acc.update(logits, 0))
    add("is_row_kernel+is_col_kernel", 2 * n * 1000 * 4, ms, "both launches of tise_is_update; logits are read twice.  LAUNCH-BOUND, not "
        "HBM-bound: two DEPENDENT launches (the column sums need every row's log-sum-exp first) move 8 MB in ~30 us -- each is a few "
        "microseconds of work behind a kernel boundary, so the HBM fraction of this entry is not a target (0.03 ms of a 38 ms device batch)")
    out["is_row_kernel+is_col_kernel"]["launch_bound"] = True
    feats = torch.rand((n, 2048), device=dev)
    sacc = device.StatsAccumulator(2048, dev)
    ms = timed(lambda: sacc.update_parts(feats, cov=False, col_sum=True))
    add("colsum_f32_sliced_kernel", n * 2048 * 4, ms, "stand-alone column sums (the product path folds them into the covariance kernel)")
    ms = timed(lambda: sacc.finalize())
    add("stats_finalize_kernel", 3 * 8 * 2048 * 2048, ms, "S, s, n -> mu, sigma (read S, write sigma; mirrors the upper tiles)")
    return out


def _self_launch(n):
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as FRESH child processes
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) and relay their output; rank 0's JSON
    line goes to this process's stdout.  Called before this process has made any GPU call -- it never initialises HIP
    and never replaces itself (no exec): it waits for the child and returns its exit code."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")              # dmabuf IPC only on this driver (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    env["TISE_BENCH_LAUNCHER"] = "self"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def fid_score_is(logits, dev):
    from tise_toolbox_amd import device
    from tise_toolbox_amd.engine import T_COCO
    acc = device.InceptionScoreAccumulator(logits.shape[1], logits.shape[0], T_COCO, 10, "coco", False, dev)
    acc.update(logits.contiguous(), 0)
    m, s, _ = acc.finalize()
    return m, s


def cross_check_fp32(eng, data, chunks, lo, n_total, mu, sigma, mu_ref, sigma_ref, fid_split, is_split, solver, dev):
    """The timed set once more through EXACT fp32 convolutions (MIOpen, immediate mode) with the same weights and
    the same statistics / Frechet / IS* kernels: what the split-fp16 operand format costs at the job's full size."""
    from tise_toolbox_amd import device
    from tise_toolbox_amd.engine import RealismEngine, T_COCO
    old = {k: os.environ.get(k) for k in ("TISE_CONV", "TISE_MIOPEN_FIND")}
    os.environ["TISE_CONV"], os.environ["TISE_MIOPEN_FIND"] = "miopen", "0"
    try:
        ref_eng = RealismEngine(dims=2048, device_index=dev.index, seed=0, with_logits=True)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    t0 = time.perf_counter()
    ref_eng.begin(n_total=n_total, temperature=T_COCO, splits=10, rule="coco")
    max_err, max_abs = torch.zeros((), device=dev), torch.zeros((), device=dev)
    sub = min(125, chunks[0][1] - chunks[0][0])               # MIOpen fp32 workspace: smaller batches than the HIP trunk
    for a, b in chunks:
        for i in range(a, b, sub):
            j = min(i + sub, b)
            f_ref = ref_eng.step_u8(data[i:j], lo + i)
            f_split, _ = eng.features_from_u8(data[i:j])
            max_err = torch.maximum(max_err, (f_split - f_ref).abs().max())
            max_abs = torch.maximum(max_abs, f_ref.abs().max())
    mu32, sig32 = ref_eng.statistics()
    fid32 = float(solver.distance(mu32, sig32, mu_ref, sigma_ref)["fid"])
    is32 = ref_eng.inception_score()
    torch.cuda.synchronize()
    return {"n": n_total, "against": "the same images and weights through MIOpen fp32 convolutions (TISE_CONV=miopen), same "
                                     "statistics / Frechet / IS* kernels",
            "fid_split": fid_split, "fid_fp32": fid32, "dfid": abs(fid_split - fid32),
            "is_split": is_split[0], "is_fp32": is32[0], "dis": abs(is_split[0] - is32[0]), "dis_std": abs(is_split[1] - is32[1]),
            "max_feature_err_rel": float((max_err / max_abs).item()), "seconds": time.perf_counter() - t0}


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException:                                              # noqa: BLE001
        # a rank that fails must END -- non-zero, at once -- so that the launcher takes the other ranks down with it instead of
        # leaving them in a collective until its timeout (dist.collective_timeout: 120 s): no interpreter shutdown, which
        # would try to destroy the process group and wait for peers
        import traceback
        traceback.print_exc()
        sys.stderr.flush()
        os._exit(1)
