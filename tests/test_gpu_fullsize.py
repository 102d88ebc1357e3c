"""The hot path's reductions at BASELINE.json's full sizes (30 000 images, d = 2048, C = 1000, 8 ranks x 3 750):
oracle comparisons where the oracle finishes in seconds, otherwise size-independent properties -- additivity over
batches and over rank shards (what the all-reduce relies on), invariance to the batching, idempotent finalize."""
import os

import numpy as np
import pytest
import torch

from oracle import fid_oracle, is_oracle, resize_oracle
from tests import _cases

pytestmark = pytest.mark.gpu

N, D, C = 30000, 2048, 1000


@pytest.fixture(scope="module")
def feats(cuda_device):
    g = torch.Generator(device="cpu").manual_seed(5)
    lat = torch.randn((N, 96), generator=g)
    mix = torch.randn((96, D), generator=g) * 0.3
    x = torch.relu(lat @ mix + 0.05 * torch.randn((N, D), generator=g)).float()      # pool3-like: non-negative, correlated
    return x.to(cuda_device)


def test_statistics_30k_additive_over_batches_and_rank_shards(cuda_device, feats):
    from tise_toolbox_amd.device import FrechetSolver, StatsAccumulator
    whole = StatsAccumulator(D, cuda_device)
    for i in range(0, N, 500):                                # bench / engine batching
        whole.update(feats[i:i + 500])
    assert whole.count() == N
    buf = whole.buffer().clone()
    # 8 rank shards of 3 750 rows, batch 50 (README recipe), buffers summed = the RCCL all-reduce
    total = torch.zeros_like(buf)
    for r in range(8):
        acc = StatsAccumulator(D, cuda_device)
        for i in range(r * 3750, (r + 1) * 3750, 50):
            acc.update(feats[i:i + 50])
        total += acc.buffer()
        acc.close()
    rel = ((total - buf).abs().max() / buf.abs().max()).item()
    assert rel <= 1e-13, rel                                   # fp64 sums: order effects only
    mu, sigma = whole.finalize()
    mu2, sigma2 = whole.finalize()
    assert torch.equal(mu, mu2) and torch.equal(sigma, sigma2)                       # finalize does not consume the sums
    assert torch.equal(sigma, sigma.t())
    # oracle at full size: np.mean / np.cov on the fp64-widened features (fid_score.py:194-195)
    x64 = feats.cpu().numpy().astype(np.float64)
    mu_o, sigma_o = np.mean(x64, axis=0), np.cov(x64, rowvar=False)
    assert np.abs(mu.cpu().numpy() - mu_o).max() <= 1e-12 * np.abs(mu_o).max()
    assert np.abs(sigma.cpu().numpy() - sigma_o).max() <= 1e-11 * np.abs(sigma_o).max()
    # Frechet distance of the set against a shifted copy of itself, statistics from the shards vs from the whole
    sh = StatsAccumulator(D, cuda_device)
    sh.buffer().copy_(total)
    mu_s, sigma_s = sh.finalize()
    solver = FrechetSolver(D, cuda_device)
    mu_b = mu + 0.01
    f1 = solver.distance(mu, sigma, mu_b, sigma)["fid"]
    f2 = solver.distance(mu_s, sigma_s, mu_b, sigma)["fid"]
    assert abs(f1 - f2) <= 1e-9
    assert abs(f1 - D * 1e-4) <= 1e-6                          # identical covariances: FID = |mu1 - mu2|^2 exactly


def test_is_star_30k_matches_oracle_and_is_shardable(cuda_device):
    from tise_toolbox_amd.device import InceptionScoreAccumulator
    g = torch.Generator(device="cpu").manual_seed(6)
    logits = (torch.randn((N, C), generator=g) * 2.0 + torch.randn((1, C), generator=g)).float()
    T = 0.9091363549232483
    want_mean, want_std = is_oracle.inception_score_from_logits(logits.numpy(), T, 10, "coco", dtype=np.float64)
    dev_logits = logits.to(cuda_device)
    acc = InceptionScoreAccumulator(C, N, T, 10, "coco", device=cuda_device)
    for i in range(0, N, 500):
        acc.update(dev_logits[i:i + 500], i)
    mean, std, scores = acc.finalize()
    assert abs(mean - want_mean) <= 1e-9 and abs(std - want_std) <= 1e-9            # budget: 1e-4
    # 8 rank shards (3 750 rows: straddle the 3 000-row split borders), sums added = all-reduce
    total = torch.zeros_like(acc.acc)
    for r in range(8):
        a = InceptionScoreAccumulator(C, N, T, 10, "coco", device=cuda_device)
        for i in range(r * 3750, (r + 1) * 3750, 250):
            a.update(dev_logits[i:i + 250], i)
        total += a.acc
    b = InceptionScoreAccumulator(C, N, T, 10, "coco", device=cuda_device)
    b.acc.copy_(total)
    mean2, std2, _ = b.finalize()
    assert abs(mean2 - mean) <= 1e-12 and abs(std2 - std) <= 1e-12


def test_resize_full_batch_bit_exact_and_batch_invariant(cuda_device):
    from tise_toolbox_amd.device import make_lut, resize_bilinear_u8
    imgs = _cases.smooth_images(500, 256, 256, seed=9)                              # one bench batch
    src = torch.from_numpy(imgs).to(cuda_device)
    out, u8 = resize_bilinear_u8(src, (299, 299), make_lut(True), channels_last=True, return_u8=True)
    for i in (0, 137, 499):                                                          # oracle (Pillow-exact) on samples
        want = resize_oracle.resize_bilinear_u8(imgs[i], 299, 299)
        assert np.array_equal(u8[i].cpu().numpy(), want)
    one = resize_bilinear_u8(src[137:138].contiguous(), (299, 299), make_lut(True), channels_last=True)
    assert torch.equal(one[0], out[137])                                             # no dependence on the batch
    # checksum of checksums: per-image byte sums of the uint8 result, batch vs per-image launches
    sums = u8.reshape(500, -1).to(torch.int64).sum(1)
    part = torch.cat([resize_bilinear_u8(src[i:i + 125].contiguous(), (299, 299), None, return_u8=True)[1].reshape(125, -1)
                      .to(torch.int64).sum(1) for i in range(0, 500, 125)])
    assert torch.equal(sums, part)


def test_bench_strong_scaling_two_ranks_on_one_gpu(cuda_device):
    """`python bench.py --gpus 2` invoked DIRECTLY (no torchrun: bench.py starts its two ranks itself, both on THIS
    GPU over a gloo rendezvous -- TISE_DIST_BACKEND / TISE_FORCE_DEVICE0): the strong-scaling job -- one image set sharded by global index, one all-reduce of the statistics, Frechet solved on
    both ranks -- must print the same scores as the one-process run of the same job, report the sharding it used and
    carry the roofline object.  (On a node the same code runs with backend nccl = RCCL.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    common = ["--steps", "4", "--warmup", "1", "--images", "2000", "--no-cpu-baseline", "--no-cross-check", "--ref-images", "2200"]
    one = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + common, cwd=root, capture_output=True, text=True, timeout=600)
    assert one.returncode == 0, one.stderr[-2000:]
    d1 = json.loads(one.stdout.strip().split("\n")[-1])
    env = dict(os.environ, TISE_DIST_BACKEND="gloo", TISE_FORCE_DEVICE0="1")
    env.pop("WORLD_SIZE", None)
    two = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"] + common,
                         cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert two.returncode == 0, two.stderr[-2000:]
    d2 = json.loads(two.stdout.strip().split("\n")[-1])
    assert d2["n_gpus"] == 2 and d2["scaling"] == "strong" and d2["steps"] == 4
    c2, c1 = d2["config"]["collective"], d1["config"]["collective"]
    assert (c2["world_size"], c2["backend"], c2["launcher"], c2["forced_one_rank_group"]) == (2, "gloo", "self", False)
    # the one-process run brings up a ONE-rank RCCL group so that the job's all-reduces really execute (round 5)
    assert (c1["world_size"], c1["backend"], c1["forced_one_rank_group"]) == (1, "nccl", True), c1
    assert d1["reference_side"]["images"] == 2200 and d2["reference_side"]["images_this_rank"] == 1100
    assert d1["config"]["step_images"] == 500 and "resident in HBM" in d1["config"]["workload"]
    assert d2["config"]["images_total"] == d1["config"]["images_total"] == 2000 and d2["config"]["images_per_gpu"] == 1000
    # 2 000 images < d = 2 048: rank-deficient covariances, where the fp64 summation order of S -- one 2 000-image device batch
    # against two ranks' 1 000-image batches (bench.rank_batch: the largest divisor <= 3 000) -- moves the distance by ~2e-7
    assert abs(d2["scores"]["fid"] - d1["scores"]["fid"]) <= 1e-6 and abs(d2["scores"]["is_mean"] - d1["scores"]["is_mean"]) <= 1e-9
    assert d2["roofline"]["frac"] > 0.05 and d2["allreduce_ms"] >= 0.0 and d2["cpu_baseline"] is None
    assert d2["value"] > 0 and abs(d2["ms_per_step"] * 4 / 1e3 * d2["value"] - 2000) < 1.0       # value = images / elapsed
