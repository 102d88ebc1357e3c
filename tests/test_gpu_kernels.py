"""GPU parity tests: every HIP entry point of libtise_hip.so against the CPU oracle, called
through the C ABI (ctypes, tise_toolbox_amd.device).  Integer work is compared bit-exactly; fp64
work with the tolerances written next to each assert."""
import os

import numpy as np
import pytest
import torch

from oracle import fid_oracle, is_oracle, resize_oracle
from tests import _cases

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev(cuda_device):
    return cuda_device


def test_device_is_gfx950(dev):
    import ctypes
    from tise_toolbox_amd import _lib
    cu, is950, mem = ctypes.c_int(), ctypes.c_int(), ctypes.c_size_t()
    _lib.call("tise_device_info", ctypes.byref(cu), ctypes.byref(is950), ctypes.byref(mem))
    assert is950.value == 1 and cu.value >= 200 and mem.value > 100 * 2 ** 30


# ------------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("m,n,k", [(64, 64, 16), (70, 130, 37), (1, 1, 1), (256, 192, 500), (33, 2048, 129)])
def test_gemm_f64_layouts(dev, m, n, k):
    from tise_toolbox_amd import device
    rng = np.random.default_rng(m * 1000 + n)
    a = rng.standard_normal((m, k))
    b = rng.standard_normal((k, n))
    ref = a @ b
    ta, tb = torch.as_tensor(a, device=dev), torch.as_tensor(b, device=dev)
    tol = 1e-13 * k * max(1.0, np.abs(ref).max())
    for A in (ta, ta.t().contiguous().t()):                 # k-contiguous and m-contiguous A
        for B in (tb, tb.t().contiguous().t()):             # n-contiguous and k-contiguous B
            c = device.gemm_f64(A, B).cpu().numpy()
            assert np.abs(c - ref).max() <= tol


def test_gemm_asymmetric_identity(dev):
    """A = I with an ASYMMETRIC B catches a transposed C write (cdna guide section 3)."""
    from tise_toolbox_amd import device
    n = 96
    b = np.arange(n * n, dtype=np.float64).reshape(n, n)
    c = device.gemm_f64(torch.eye(n, dtype=torch.float64, device=dev), torch.as_tensor(b, device=dev)).cpu().numpy()
    np.testing.assert_array_equal(c, b)


# ------------------------------------------------------------------------------------------- resize
def test_resize_golden_bit_exact(dev, golden_dir):
    from tise_toolbox_amd import device
    g = np.load(os.path.join(golden_dir, "pil_resize_299.npz"))
    names = [k[3:] for k in g.files if k.startswith("in_")]
    lut = device.make_lut(True)
    for k in names:
        src = torch.as_tensor(g["in_" + k], device=dev).unsqueeze(0)
        for cl in (True, False):
            out, u8 = device.resize_bilinear_u8(src, (299, 299), lut, channels_last=cl, return_u8=True)
            np.testing.assert_array_equal(u8[0].cpu().numpy(), g["out_" + k], err_msg=f"{k} cl={cl}")
            # fused ToTensor + inception.py:120-124 affine: bit-exact against the numpy restatement
            want = resize_oracle.normalize_input(resize_oracle.to_tensor(g["out_" + k]))
            got = out[0].cpu().numpy()
            assert got.shape == (3, 299, 299)
            np.testing.assert_array_equal(got, want, err_msg=f"{k} cl={cl} float")
            assert out.is_contiguous(memory_format=torch.channels_last if cl else torch.contiguous_format)


@pytest.mark.parametrize("h,w", [(256, 256), (64, 48), (299, 299), (300, 299), (299, 301), (517, 31), (1024, 768), (7, 5)])
def test_resize_random_sizes_vs_oracle(dev, h, w):
    from tise_toolbox_amd import device
    rng = np.random.default_rng(h * 7 + w)
    n = 3
    imgs = rng.integers(0, 256, (n, h, w, 3), dtype=np.uint8)
    out, u8 = device.resize_bilinear_u8(torch.as_tensor(imgs, device=dev), (299, 299), device.make_lut(False),
                                        channels_last=True, return_u8=True)
    only = device.resize_u8_only(torch.as_tensor(imgs, device=dev), (299, 299))     # the product path: uint8 out only (4 bytes per lane)
    for i in range(n):
        want = resize_oracle.resize_bilinear_u8(imgs[i], 299, 299)
        np.testing.assert_array_equal(u8[i].cpu().numpy(), want)
        np.testing.assert_array_equal(only[i].cpu().numpy(), want)
        np.testing.assert_array_equal(out[i].cpu().numpy(), resize_oracle.to_tensor(want))
    # a destination that is not a multiple of four bytes wide and not 299: the dword stores' row tails
    for (oh, ow) in ((61, 37), (300, 298)):
        got = device.resize_u8_only(torch.as_tensor(imgs[:1], device=dev), (oh, ow))
        np.testing.assert_array_equal(got[0].cpu().numpy(), resize_oracle.resize_bilinear_u8(imgs[0], oh, ow))


def test_resize_batch_and_empty(dev):
    from tise_toolbox_amd import device
    imgs = _cases.smooth_images(5, 256, 256, seed=1)
    out, u8 = device.resize_bilinear_u8(torch.as_tensor(imgs, device=dev), return_u8=True)
    for i in range(5):
        np.testing.assert_array_equal(u8[i].cpu().numpy(), resize_oracle.resize_bilinear_u8(imgs[i], 299, 299))
    empty = device.resize_bilinear_u8(torch.empty((0, 256, 256, 3), dtype=torch.uint8, device=dev))
    assert tuple(empty.shape) == (0, 3, 299, 299)


# ------------------------------------------------------------------------------------------- statistics
@pytest.mark.parametrize("n,d,chunks", [(35, 24, [35]), (300, 64, [100, 7, 193]), (500, 192, [250, 250]),
                                        (257, 2048, [50, 50, 50, 107]), (64, 768, [1, 63])])
def test_stats_match_np_cov(dev, n, d, chunks):
    from tise_toolbox_amd import device
    x = _cases.pool3_like_features(n, d, seed=n + d)
    acc = device.StatsAccumulator(d, dev)
    lo = 0
    for c in chunks:
        acc.update(torch.as_tensor(x[lo:lo + c], device=dev))
        lo += c
    assert acc.count() == n
    mu, sigma = acc.finalize()
    mu_ref, sigma_ref = fid_oracle.calculate_activation_statistics(x)       # np.mean / np.cov on float64
    scale = np.abs(sigma_ref).max()
    assert np.abs(mu.cpu().numpy() - mu_ref).max() <= 1e-14 * max(1.0, np.abs(mu_ref).max())
    # S - n mu mu^T cancels ~1 digit; 1e-12 relative to the largest covariance entry
    assert np.abs(sigma.cpu().numpy() - sigma_ref).max() <= 1e-12 * scale
    s = sigma.cpu().numpy()
    np.testing.assert_array_equal(s, s.T)                                    # mirrored tiles: exactly symmetric


def test_stats_golden_fake_model(dev, golden_dir):
    """G2: features produced by the reference's own get_activations loop -> its mu/sigma."""
    from tise_toolbox_amd import device
    g = np.load(os.path.join(golden_dir, "actstats_fake_model.npz"))
    feats32 = g["feats32"]
    np.testing.assert_array_equal(feats32.astype(np.float64), g["act"])      # fp32 -> fp64 widening is exact
    acc = device.StatsAccumulator(feats32.shape[1], dev)
    bs = int(g["batch_size"])
    for i in range(0, feats32.shape[0], bs):
        acc.update(torch.as_tensor(feats32[i:i + bs], device=dev))
    mu, sigma = acc.finalize()
    np.testing.assert_allclose(mu.cpu().numpy(), g["mu"], rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(sigma.cpu().numpy(), g["sigma"], rtol=0, atol=1e-12 * np.abs(g["sigma"]).max())


def test_stats_strided_rows_reset_and_merge(dev):
    from tise_toolbox_amd import device
    d = 192
    x = _cases.pool3_like_features(400, d, seed=5)
    wide = torch.zeros((400, d + 64), dtype=torch.float32, device=dev)
    wide[:, :d] = torch.as_tensor(x, device=dev)
    a = device.StatsAccumulator(d, dev)
    a.update(wide[:, :d])                                      # ld > d
    b1, b2 = device.StatsAccumulator(d, dev), device.StatsAccumulator(d, dev)
    b1.update(torch.as_tensor(x[:150], device=dev))
    b2.update(torch.as_tensor(x[150:], device=dev))
    merged = b1.buffer() + b2.buffer()                         # what the all-reduce computes
    np.testing.assert_allclose(merged.cpu().numpy(), a.buffer().cpu().numpy(), rtol=1e-13, atol=1e-13)
    a.reset()
    assert a.count() == 0 and float(a.buffer().abs().max()) == 0.0
    a.update(torch.empty((0, d), dtype=torch.float32, device=dev))   # empty batch is a no-op
    assert a.count() == 0


# ------------------------------------------------------------------------------------------- linear algebra pieces
@pytest.mark.parametrize("d,kind", [(8, "fullrank"), (64, "fullrank"), (64, "rankdef"), (192, "rankdef"), (300, "fullrank")])
def test_pivoted_cholesky_reconstructs(dev, d, kind):
    from tise_toolbox_amd import device
    _, s1, _, _ = _cases.frechet_case(d, kind, seed=d)
    solver = device.FrechetSolver(d, dev)
    lt, r = solver.pivoted_cholesky(s1)
    lt = lt.cpu().numpy()
    rank_ref = np.linalg.matrix_rank(s1)
    assert rank_ref <= r <= d
    if kind == "fullrank":
        assert r == d
    assert np.abs(lt[r:]).max() == 0.0 if r < d else True
    rec = lt.T @ lt
    assert np.abs(rec - s1).max() <= 1e-13 * d * np.abs(s1).max()


@pytest.mark.parametrize("n", [1, 2, 3, 17, 64, 65, 200, 515, 1030, 2048])
def test_eigvalsh_matches_lapack(dev, n):
    from tise_toolbox_amd import device
    rng = np.random.default_rng(n)
    a = rng.standard_normal((n, n))
    a = a @ a.T / n + np.diag(rng.uniform(0, 1, n))
    if n > 16:                                           # clustered + zero eigenvalues
        q, _ = np.linalg.qr(rng.standard_normal((n, n)))
        lam = np.concatenate([np.zeros(n // 3), np.full(n // 3, 0.5), rng.uniform(0, 3, n - 2 * (n // 3))])
        a = (q * lam) @ q.T
        a = (a + a.T) / 2
    solver = device.FrechetSolver(max(n, 8), dev)
    w = solver.eigvalsh(torch.as_tensor(a, device=dev)).cpu().numpy()
    ref = np.linalg.eigvalsh(a)
    assert np.all(np.diff(w) >= -1e-300)                 # ascending
    assert np.abs(w - ref).max() <= 5e-14 * n * max(1.0, np.abs(ref).max())


def _clustered_spd(n, seed):
    rng = np.random.default_rng(seed)
    q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    lam = np.concatenate([np.zeros(n // 3), np.full(n // 3, 0.5), rng.uniform(0, 3, n - 2 * (n // 3))])
    a = (q * lam) @ q.T
    return (a + a.T) / 2


def test_eigvalsh_paths_beyond_the_metric_size(dev, monkeypatch):
    """VERDICT r3 (weak 8): tise_stats_create / tise_frechet_create accept d <= 8192, but the suite stopped at the metric's
    2048, where sytrd_fused8_kernel serves every column.  The other tridiagonalisation schemes: n = 2304 (> 2048: the
    generic one-launch-per-column kernel, sytrd_fused_kernel, whose three vectors live in LDS up to n = 6144), the same
    kernel forced at n = 300 (TISE_SYTRD_FUSED_GENERIC), and -- in a fresh process, the switch is read once --
    round 1's two-launch scheme (sytrd_update_matvec_kernel + sytrd_step_kernel: n > 6144) forced at n = 300."""
    import subprocess
    import sys
    from tise_toolbox_amd import device
    for n, env in ((2304, None), (300, "TISE_SYTRD_FUSED_GENERIC")):
        if env:
            monkeypatch.setenv(env, "1")
        a = _clustered_spd(n, n)
        w = device.FrechetSolver(n, dev).eigvalsh(torch.as_tensor(a, device=dev)).cpu().numpy()
        ref = np.linalg.eigvalsh(a)
        assert np.all(np.diff(w) >= -1e-300)
        assert np.abs(w - ref).max() <= 5e-14 * n * max(1.0, np.abs(ref).max()), (n, env)
        if env:
            monkeypatch.delenv(env)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import numpy as np, torch, sys\n"
            "sys.path.insert(0, %r)\n"
            "from tests.test_gpu_kernels import _clustered_spd\n"
            "from tise_toolbox_amd import device\n"
            "a = _clustered_spd(300, 7)\n"
            "w = device.FrechetSolver(300, torch.device('cuda', 0)).eigvalsh(torch.as_tensor(a, device='cuda:0')).cpu().numpy()\n"
            "err = np.abs(w - np.linalg.eigvalsh(a)).max()\n"
            "print('two-launch err', err)\n"
            "assert err <= 5e-14 * 300 * 3.0\n") % root
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, TISE_SYTRD_TWO_LAUNCH="1", PYTHONPATH=root),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------- Frechet distance
@pytest.mark.parametrize("d", [8, 64, 192])
@pytest.mark.parametrize("kind", ["fullrank", "rankdef", "identical", "shifted"])
def test_frechet_golden(dev, golden_dir, d, kind):
    """G1: inputs + the reference's own calculate_frechet_distance output."""
    from tise_toolbox_amd import fid_score
    g = np.load(os.path.join(golden_dir, f"frechet_d{d}_{kind}.npz"))
    got = fid_score.calculate_frechet_distance(g["mu1"], g["sigma1"], g["mu2"], g["sigma2"])
    assert isinstance(got, np.float64)
    # north_star budget is |dFID| <= 1e-3; the device path is expected far inside it
    assert abs(got - float(g["fid"])) <= 2e-5, (got, float(g["fid"]))
    if kind == "fullrank":
        assert abs(got - float(g["fid"])) <= 1e-9
    res = fid_score.calculate_frechet_distance.last_result
    assert res["flags"] & 1 == 0
    assert (res["rank"] < d) == (kind == "rankdef")


@pytest.mark.parametrize("kind", ["fullrank", "rankdef"])
def test_frechet_d2048_vs_reference_scalar(dev, golden_dir, kind):
    """G5: BASELINE's d = 2048, incl. the N = 1000 < d rank-deficient case of config 1."""
    from tise_toolbox_amd import fid_score
    g = np.load(os.path.join(golden_dir, f"frechet_d2048_{kind}.npz"))
    mu1, s1, mu2, s2 = _cases.frechet_case_2048(kind, int(g["n1"]), int(g["n2"]))
    got = fid_score.calculate_frechet_distance(mu1, s1, mu2, s2)
    ref = float(g["fid"])
    assert abs(got - ref) <= 1e-3                        # the stated tolerance
    assert abs(got - ref) <= (1e-9 if kind == "fullrank" else 1e-4), (got, ref)
    res = fid_score.calculate_frechet_distance.last_result
    assert res["rank"] == 2048 if kind == "fullrank" else res["rank"] < 1100


@pytest.mark.parametrize("kind", ["fullrank", "rankdef"])
def test_frechet_prefactored_equals_one_call_form(dev, kind):
    """tise_frechet_prefactor (side stream) + tise_frechet_distance_prefactored == tise_frechet_distance, bit for bit
    (same kernels, the Cholesky merely hoisted); symmetric use: the FACTORED side may be either argument."""
    from tise_toolbox_amd import device
    d = 192
    m1, s1, m2, s2 = _cases.frechet_case(d, kind, seed=5)
    solver = device.FrechetSolver(d, dev)
    one = solver.distance(m1, s1, m2, s2)
    solver.prefactor(torch.as_tensor(s1, device=dev))
    two = solver.distance_prefactored(m1, m2, s2)
    assert two["fid"] == one["fid"] and two["rank"] == one["rank"] and two["flags"] == one["flags"]
    assert solver.prefactor_ms() > 0.0
    want = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    solver.prefactor(torch.as_tensor(s2, device=dev))                  # factor the OTHER side
    three = solver.distance_prefactored(m2, m1, s1)
    assert abs(three["fid"] - want) <= (1e-9 if kind == "fullrank" else 2e-5)
    fresh = device.FrechetSolver(d, dev)
    with pytest.raises(Exception):
        fresh.distance_prefactored(m1, m2, s2)                          # nothing factored yet


@pytest.mark.parametrize("d", [100, 300, 1000])
@pytest.mark.parametrize("pivoted", [False, True])
def test_frechet_partial_last_panel_both_factorisations(dev, d, pivoted, monkeypatch):
    """the unpivoted 64-column Cholesky with a PARTIAL last panel (d % 64 = 36, 44, 40) through
    tise_frechet_distance, against the scipy-sqrtm oracle, and the same inputs through the pivoted factorisation (the
    TISE_CHOL_PIVOTED switch is read per call): both within 1e-9 (full rank) and of each other."""
    from tise_toolbox_amd import device
    m1, s1, m2, s2 = _cases.frechet_case(d, "fullrank", seed=11)
    want = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    solver = device.FrechetSolver(d, dev)
    if pivoted:
        monkeypatch.setenv("TISE_CHOL_PIVOTED", "1")
    got = solver.distance(m1, s1, m2, s2)
    assert got["rank"] == d and abs(got["fid"] - want) <= 1e-9 * max(1.0, abs(want)), (got["fid"], want)
    solver.set_profiling(True)

    def factor_ms():                                               # best of three: a single timed call can catch a hiccup of the box
        best = float("inf")
        for _ in range(3):
            solver.distance(m1, s1, m2, s2)
            best = min(best, solver.phase_ms()["pchol"])
        return best
    t_sel = factor_ms()
    monkeypatch.delenv("TISE_CHOL_PIVOTED", raising=False)
    t_unpivoted = factor_ms()
    if pivoted and d >= 300:
        assert t_sel > t_unpivoted                                 # the switch really selected the slower, pivoted kernels


def test_frechet_lost_prefactor_is_detected(dev):
    """tise_pivoted_cholesky overwrites the solver's factor buffer, so a factor left by prefactor() must be
    invalidated (the prefactored distance then fails instead of silently using the wrong factor), and
    fid_score.calculate_fid_given_paths's fall-back path -- the one-call form -- gives the right answer."""
    from tise_toolbox_amd import _lib, device
    d = 192
    m1, s1, m2, s2 = _cases.frechet_case(d, "fullrank", seed=5)
    other = _cases.frechet_case(d, "fullrank", seed=6)[1]
    solver = device.FrechetSolver(d, dev)
    solver.prefactor(torch.as_tensor(s1, device=dev))
    solver.pivoted_cholesky(torch.as_tensor(other, device=dev))            # overwrites h->lt
    with pytest.raises(_lib.TiseStatusError):
        solver.distance_prefactored(m1, m2, s2)
    want = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    assert abs(solver.distance(m1, s1, m2, s2)["fid"] - want) <= 1e-9


def test_frechet_properties(dev):
    from tise_toolbox_amd import fid_score
    d = 128
    x = _cases.pool3_like_features(700, d, seed=9)
    y = _cases.pool3_like_features(650, d, seed=10, shift=0.1)
    m1, s1 = _cases.stats(x)
    m2, s2 = _cases.stats(y)
    f12 = fid_score.calculate_frechet_distance(m1, s1, m2, s2)
    f21 = fid_score.calculate_frechet_distance(m2, s2, m1, s1)
    assert abs(f12 - f21) <= 1e-9 * max(1.0, f12)        # symmetric in its arguments
    f11 = fid_score.calculate_frechet_distance(m1, s1, m1, s1)
    assert abs(f11) <= 1e-9                              # FID(X, X) ~ 0 (may be slightly negative: no clamp)
    perm = np.random.default_rng(0).permutation(700)     # permutation invariance of the statistics
    m1p, s1p = _cases.stats(x[perm])
    assert abs(fid_score.calculate_frechet_distance(m1p, s1p, m2, s2) - f12) <= 1e-9
    assert abs(f12 - fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)) <= 1e-9


def test_frechet_errors_and_nonfinite(dev, capsys):
    from tise_toolbox_amd import fid_score
    with pytest.raises(AssertionError):
        fid_score.calculate_frechet_distance(np.zeros(3), np.eye(3), np.zeros(4), np.eye(4))
    with pytest.raises(AssertionError):
        fid_score.calculate_frechet_distance(np.zeros(3), np.eye(3), np.zeros(3), np.eye(4))
    s = np.eye(4)
    s[1, 1] = np.nan
    out = fid_score.calculate_frechet_distance(np.zeros(4), s, np.zeros(4), np.eye(4))
    assert "adding 1e-06 to diagonal of cov estimates" in capsys.readouterr().out      # fid_score.py:157-158
    assert not np.isfinite(out)
    # scalars / 1-d input go through atleast_1d / atleast_2d like the reference (:143-147)
    assert abs(fid_score.calculate_frechet_distance(1.0, 4.0, 3.0, 9.0) - (4.0 + 4.0 + 9.0 - 2 * 6.0)) <= 1e-12


# ------------------------------------------------------------------------------------------- IS*
@pytest.mark.parametrize("name", ["coco", "ois", "bird"])
def test_is_golden(dev, golden_dir, name):
    from tise_toolbox_amd import inception_score as isc
    g = np.load(os.path.join(golden_dir, f"is_reduce_{name}.npz"))
    rule = "ois" if str(g["rule"]) == "ois" else "coco"
    mean, std, scores = isc.inception_score_from_logits(g["logits"], float(g["temperature"]), int(g["splits"]), rule,
                                                        bool(g["drop_first"]), return_scores=True)
    # |dIS| <= 1e-4 is the north_star budget against the reference arithmetic (fp32 for coco/bird)
    assert abs(mean - float(g["mean32"])) <= 1e-4 and abs(std - float(g["std32"])) <= 1e-4
    # against exact (fp64) evaluation of the same formula the kernel is at rounding level
    assert abs(mean - float(g["mean64"])) <= 1e-11 and abs(std - float(g["std64"])) <= 1e-11
    assert np.all(scores >= 1.0 - 1e-12) and np.all(scores <= g["logits"].shape[1])    # 1 <= IS <= C


@pytest.mark.parametrize("name", ["is_ref_coco_57.npz", "is_ref_coco_130.npz", "is_ref_bird_150.npz", "is_ref_ois_97.npz"])
def test_is_matches_reference_script_run(dev, golden_dir, name):
    """is_score.hip against what the REFERENCE scripts computed from the same logits when run by path under
    stub tensorflow / torchvision (tests/golden/make_golden_is.py).  |dIS| <= 1e-4 (north_star)."""
    from tise_toolbox_amd import inception_score as isc
    g = np.load(os.path.join(golden_dir, name))
    rule, drop = str(g["rule"]), bool(g["drop_first"])
    T = is_oracle.T_OIS if rule == "ois" else float(g["temperature"])
    mean, std = isc.inception_score_from_logits(g["logits"], T, int(g["splits"]), rule, drop)
    assert abs(mean - float(g["mean"])) <= 1e-4 and abs(std - float(g["std"])) <= 1e-4, (mean, std, g["mean"], g["std"])
    if rule == "coco" and not drop:
        assert "[Inception Score] mean: {:.5f} std: {:.5f}".format(mean, std) == str(g["expected_text"])


@pytest.mark.parametrize("n,c,splits,rule", [(1000, 1000, 10, "coco"), (997, 80, 10, "ois"), (23, 51, 10, "coco"),
                                             (10, 7, 10, "coco"), (64, 1008, 3, "coco")])
def test_is_sharded_updates_match_oracle(dev, n, c, splits, rule):
    """Rows arrive in ragged chunks that straddle split borders (the data-parallel case)."""
    from tise_toolbox_amd import device
    rng = np.random.default_rng(n + c)
    logits = (rng.standard_normal((n, c)) * 2.5).astype(np.float32)
    T = is_oracle.T_OIS if rule == "ois" else is_oracle.T_COCO
    acc = device.InceptionScoreAccumulator(c, n, T, splits, rule, False, dev)
    cuts = sorted(set([0, n] + list(rng.integers(0, n + 1, 4))))
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        acc.update(torch.as_tensor(logits[lo:hi], device=dev), lo)
    mean, std, _ = acc.finalize()
    m64, s64 = is_oracle.inception_score_from_logits(logits, T, splits, rule, dtype=np.float64)
    assert abs(mean - m64) <= 1e-10 and abs(std - s64) <= 1e-10
    A, B = is_oracle.is_sums(logits, T, 0, n, splits, rule)
    got = acc.acc.cpu().numpy()
    np.testing.assert_allclose(got[:splits], A, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(got[splits:].reshape(splits, c), B, rtol=1e-12, atol=1e-13)


# ------------------------------------------------------------------------------------------- trunk epilogues
def _nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous()


def test_bias_relu_slices_exact(dev):
    """Reads a channel slice of a wider raw tensor, writes a channel slice of a concat buffer; exact."""
    from tise_toolbox_amd.trunk import FusedTrunk
    g = torch.Generator(device="cpu").manual_seed(0)
    raw = torch.randn((3, 7, 5, 176), generator=g).to(dev)
    bias = torch.randn(176, generator=g).to(dev)
    out = torch.full((3, 7, 5, 256), -7.0, device=dev)
    FusedTrunk._bias_relu(raw, bias[64:112], 64, 48, out, 128)
    want = torch.relu(raw[..., 64:112] + bias[64:112])
    assert torch.equal(out[..., 128:176], want)
    assert torch.all(out[..., :128] == -7.0) and torch.all(out[..., 176:] == -7.0)      # nothing else touched
    packed = FusedTrunk._bias_relu(raw, bias[112:176], 112, 64)
    assert packed.shape == (3, 7, 5, 64) and torch.equal(packed, torch.relu(raw[..., 112:176] + bias[112:176]))
    r2 = raw.clone()
    same = FusedTrunk._bias_relu(r2, bias)                                                # in place
    assert same.data_ptr() == r2.data_ptr() and torch.equal(r2, torch.relu(raw + bias))


def test_avgpool_and_maxpool_vs_torch(dev):
    import torch.nn.functional as F
    from tise_toolbox_amd.trunk import FusedTrunk
    g = torch.Generator(device="cpu").manual_seed(1)
    raw = torch.randn((2, 9, 11, 96), generator=g).to(dev)
    bias = torch.randn(32, generator=g).to(dev)
    out = torch.zeros((2, 9, 11, 64), device=dev)
    FusedTrunk._avgpool_bias_relu(raw, bias, 48, 32, out, 16)
    x = raw[..., 48:80].permute(0, 3, 1, 2)
    want = torch.relu(F.avg_pool2d(x, 3, 1, 1) + bias.view(1, -1, 1, 1)).permute(0, 2, 3, 1)     # count_include_pad=True
    assert (out[..., 16:48] - want).abs().max().item() <= 1e-6
    xin = torch.randn((2, 13, 9, 48), generator=g).to(dev)
    b48 = torch.randn(48, generator=g).to(dev)
    got = FusedTrunk._maxpool(xin, b48)
    want = F.max_pool2d(torch.relu(xin.permute(0, 3, 1, 2) + b48.view(1, -1, 1, 1)), 3, 2).permute(0, 2, 3, 1)
    assert got.shape == (2, 6, 4, 48) and torch.equal(got, want.contiguous())
    cat = torch.zeros((2, 6, 4, 64), device=dev)
    FusedTrunk._maxpool(xin, None, cat, 16)
    assert torch.equal(cat[..., 16:], F.max_pool2d(xin.permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1))


@pytest.mark.parametrize("dims", [2048, 768, 192, 64])
def test_fused_trunk_matches_module_graph(dev, dims):
    """MIOpen convs + fused HIP epilogues (1x1 fusion, pool/conv commutation, no cat) vs the plain
    torchvision-order module graph on the same device and weights."""
    from tise_toolbox_amd.inception import InceptionV3
    from tise_toolbox_amd.trunk import FusedTrunk
    torch.backends.cudnn.benchmark = False
    m = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]], seed=0).to(dev).eval()
    x = torch.rand((6, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = m(x, prenormalized=True)[0]
        got = FusedTrunk(m, dev)(x)
    assert got.shape == want.shape
    err = (got - want).abs().max().item()
    assert err <= 2e-4 * want.abs().max().item(), err


# ------------------------------------------------------------------------------------------- split-precision conv
@pytest.mark.parametrize("shape", [(17, 17, 160, 160, 1, 7, 1, (0, 3)), (35, 35, 48, 64, 5, 5, 1, (2, 2)),
                                   (35, 35, 288, 384, 3, 3, 2, (0, 0)), (9, 9, 80, 192, 3, 3, 1, (0, 0)),
                                   (8, 8, 320, 1344, 1, 1, 1, (0, 0)), (11, 7, 32, 48, 3, 3, 1, (1, 1))])
@pytest.mark.parametrize("variant", ["fast", "glds"])
def test_conv_split_matches_fp64_conv(dev, shape, variant):
    """3-term split fp16 MFMA conv vs an fp64 convolution: fp32-class accuracy (|err| <= 4e-6 of the
    output scale), including M/N/K tails, padding, stride, channel counts with Cin % 32 == 16 (input AND
    destination tensors with a 16-channel tail block), for the default and the generic kernel."""
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    H, W, Cin, Cout, kh, kw, st, pad = shape
    g = torch.Generator(device="cpu").manual_seed(Cin + Cout)
    n = 5
    x = (torch.rand((n, H, W, Cin), generator=g) * 3.0).to(dev)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (st, st), pad, dev, variant=variant)
    oh, ow = conv.out_hw(H, W)
    out = torch.zeros((n, oh, ow, 2 * (Cout + 32)), dtype=torch.float16, device=dev)
    raw = torch.zeros((n, oh, ow, 16), dtype=torch.float32, device=dev)
    # three destinations: a shifted slice of a wider tensor, a raw fp32 slice, the rest
    conv(split(x), [(0, 16, out, 16, 0), (16, 32, raw, 0, 1), (32, Cout, out, 64, 0)])
    ref_lin = torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, st, pad).permute(0, 2, 3, 1)
    ref = torch.relu(ref_lin + b.double())
    scale = ref.abs().max().item()
    got = merge(out)
    assert (got[..., 16:32].double() - ref[..., 0:16]).abs().max().item() <= 4e-6 * scale
    assert (got[..., 64:].double() - ref[..., 32:]).abs().max().item() <= 4e-6 * scale
    assert (raw.double() - ref_lin[..., 16:32]).abs().max().item() <= 4e-6 * scale
    assert got[..., :16].abs().max().item() == 0 and got[..., 32:64].abs().max().item() == 0     # untouched


def test_split_pool_ops(dev):
    import torch.nn.functional as F
    from tise_toolbox_amd.conv_split import merge, split
    from tise_toolbox_amd.trunk import SplitTrunk
    g = torch.Generator(device="cpu").manual_seed(3)
    x = (torch.rand((2, 13, 9, 48), generator=g) * 5).to(dev)
    xs = split(x)
    assert (merge(xs) - x).abs().max().item() <= 2 ** -21 * 5            # 22-bit representation
    got = SplitTrunk._maxpool_split(xs)
    want = F.max_pool2d(merge(xs).permute(0, 3, 1, 2), 3, 2).permute(0, 2, 3, 1)
    assert torch.equal(merge(got), want.contiguous())
    # average-pool tail (column-walking kernel): the trunk's shapes, one-pixel-wide / one-row images, a slice in the middle
    for (n, h, w, c, co, off) in [(2, 9, 11, 32, 64, 16), (3, 8, 8, 192, 2048, 1856), (2, 35, 35, 64, 288, 224), (5, 17, 17, 192, 768, 576),
                                  (2, 1, 6, 16, 48, 32), (2, 7, 1, 8, 16, 8)]:
        raw = torch.randn((n, h, w, c), generator=g).to(dev)
        bias = torch.randn(c, generator=g).to(dev)
        out = torch.zeros((n, h, w, 2 * co), dtype=torch.float16, device=dev)
        SplitTrunk._avgpool_split(raw, bias, out, off)
        want = torch.relu(F.avg_pool2d(raw.permute(0, 3, 1, 2), 3, 1, 1) + bias.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
        got = merge(out)
        assert (got[..., off:off + c] - want).abs().max().item() <= 2e-6 * want.abs().max().item(), (n, h, w, c)
        assert not torch.cat([got[..., :off], got[..., off + c:]], -1).any()


@pytest.mark.parametrize("dims", [2048, 768, 192, 64])
def test_split_trunk_matches_module_graph(dev, dims, monkeypatch):
    """The all-HIP trunk at every --dims of the reference (inception.py:14-19; the map of blocks 0-2 averaged as
    fid_score.py:110-111 does) against the fp32 module graph; with the stem max-pools fused into their consumers and as
    separate kernels; and through the engine from uint8 images (the path the CLIs take)."""
    import torch.nn.functional as F
    from tise_toolbox_amd.engine import RealismEngine
    from tise_toolbox_amd.inception import InceptionV3
    from tise_toolbox_amd.trunk import SplitTrunk
    torch.backends.cudnn.benchmark = False
    m = InceptionV3([InceptionV3.BLOCK_INDEX_BY_DIM[dims]], seed=0).to(dev).eval()
    x = torch.rand((6, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        want = F.adaptive_avg_pool2d(m(x, prenormalized=True)[0], (1, 1))
        got = SplitTrunk(m, dev)(x)
        monkeypatch.setenv("TISE_POOL_FUSE", "0")
        got_sep = SplitTrunk(m, dev)(x)
        monkeypatch.delenv("TISE_POOL_FUSE")
    assert got.shape == want.shape == (6, dims, 1, 1)
    err = (got - want).abs().max().item()
    assert err <= 2e-4 * want.abs().max().item(), err
    assert torch.equal(got, got_sep)
    # ... and against the CPU ORACLE (not only the GPU module graph): the same block output of oracle/inception_oracle.py
    # (unfolded BatchNorm, CPU fp32, no code shared with the product module), averaged as fid_score.py:110-111 does
    from oracle import inception_oracle
    from tise_toolbox_amd.inception import build_inception3
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}    # the same seeded parameters, torchvision keys
    blk = InceptionV3.BLOCK_INDEX_BY_DIM[dims]
    o = inception_oracle.inception_forward(sd, x.cpu().contiguous(), last_block=blk, resize_input=False,
                                           normalize_input=False)[blk]          # x is already the network input
    o = o.mean(dim=(2, 3), keepdim=True) if o.shape[2] != 1 else o
    assert o.shape == got.shape
    err_o = (got.cpu() - o).abs().max().item()
    print(f"--dims {dims}: HIP trunk vs CPU oracle max err {err_o:.3e} of scale {o.abs().max().item():.3e}")
    assert err_o <= 2e-4 * o.abs().max().item(), err_o
    eng = RealismEngine(dims=dims, seed=0)
    assert isinstance(eng.fused, SplitTrunk)
    u8 = torch.randint(0, 256, (5, 64, 48, 3), dtype=torch.uint8, device=dev)
    feats, _ = eng.features_from_u8(u8)
    assert tuple(feats.shape) == (5, dims)
    monkeypatch.setenv("TISE_CONV", "miopen")
    ref_eng = RealismEngine(dims=dims, seed=0)
    assert not isinstance(ref_eng.fused, SplitTrunk)
    ref, _ = ref_eng.features_from_u8(u8)
    assert (feats - ref).abs().max().item() <= 2e-4 * ref.abs().max().item()


@pytest.mark.parametrize("variant", ["glds", "fast"])
def test_conv_split_variants_bitwise_identical_and_repeatable(dev, variant):
    """Both kernels implement the same arithmetic in the same order when Cin % 32 == 0 (two independent addressing
    schemes, LDS layouts and weight packings): outputs must be bit-identical to the generic kernel's, run after run
    (a DMA/LDS race would show up as a mismatch)."""
    from tise_toolbox_amd.conv_split import SplitConv, split
    g = torch.Generator(device="cpu").manual_seed(7)
    for (n, H, W, Cin, Cout, kh, kw, st, pad) in [(37, 17, 17, 128, 192, 7, 1, 1, (3, 0)), (3, 35, 35, 64, 96, 3, 3, 1, (1, 1)),
                                                  (64, 8, 8, 448, 384, 3, 3, 1, (1, 1)), (1, 9, 9, 32, 64, 3, 3, 2, (0, 0))]:
        x = (torch.rand((n, H, W, Cin), generator=g) * 2.0).to(dev)
        w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        xs = split(x)
        gen = SplitConv(w, b, (st, st), pad, dev, variant="glds")
        oh, ow = gen.out_hw(H, W)
        ref = torch.zeros((n, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
        gen(xs, [(0, Cout, ref, 0, 0)])
        conv = SplitConv(w, b, (st, st), pad, dev, variant=variant)
        for rep in range(6):
            out = torch.full_like(ref, 7.0)
            conv(xs, [(0, Cout, out, 0, 0)])
            assert torch.equal(out, ref), (variant, rep, (n, H, W, Cin, Cout))


@pytest.mark.parametrize("case", [(17, 17, 192, 192, 1, 7, (0, 3), 3, 37), (35, 35, 96, 96, 3, 3, (1, 1), 3, 9), (8, 8, 448, 384, 3, 3, (1, 1), 4, 70),
                                  (8, 8, 384, 384, 1, 3, (0, 1), 4, 33), (147, 147, 32, 64, 3, 3, (1, 1), 2, 3), (9, 13, 64, 80, 5, 3, (2, 1), 3, 21),
                                  (6, 5, 32, 48, 1, 3, (0, 1), 2, 50), (73, 73, 80, 192, 3, 3, (0, 0), 3, 2), (35, 35, 48, 64, 5, 5, (2, 2), 2, 7),
                                  (12, 10, 80, 96, 2, 4, (1, 0), 3, 19), (9, 9, 112, 64, 3, 3, (0, 1), 2, 23)])
def test_conv_rowwin_kernel_matches_fp64_conv(dev, case):
    """Row-window kernel (the kw taps of a filter row share ONE fetch of the pixel operand, K order (kh, block, kw)):
    against an fp64 convolution at the default kernel's tolerance -- image-row and image boundaries inside tiles (the
    zero rows of the window), top / bottom padding (zero-page lines per kh), valid and asymmetric padding, Cin = 32 n + 16
    (tail groups pairing two taps per step, odd and even KW), M and Cout tails, three destination segments incl. raw
    fp32, rows shorter than 8 pixels -- and bit-identical over repeated runs."""
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    H, W, Cin, Cout, kh, kw, pad, tn, n = case
    g = torch.Generator(device="cpu").manual_seed(H + Cin + Cout + kw)
    x = (torch.rand((n, H, W, Cin), generator=g) * 3.0).to(dev)
    w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (1, 1), pad, dev, tn=tn, variant="rowwin")
    oh, ow = conv.out_hw(H, W)
    ref_lin = torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, pad).permute(0, 2, 3, 1)
    ref = torch.relu(ref_lin + b.double())
    scale = ref.abs().max().item()
    xs = split(x)
    first = None
    for rep in range(3):
        out = torch.zeros((n, oh, ow, 2 * (Cout + 32)), dtype=torch.float16, device=dev)
        raw = torch.zeros((n, oh, ow, 16), dtype=torch.float32, device=dev)
        conv(xs, [(0, 16, out, 16, 0), (16, 32, raw, 0, 1), (32, Cout, out, 64, 0)])
        got = merge(out)
        assert (got[..., 16:32].double() - ref[..., 0:16]).abs().max().item() <= 4e-6 * scale
        assert (got[..., 64:].double() - ref[..., 32:]).abs().max().item() <= 4e-6 * scale
        assert (raw.double() - ref_lin[..., 16:32]).abs().max().item() <= 4e-6 * scale
        assert got[..., :16].abs().max().item() == 0 and got[..., 32:64].abs().max().item() == 0
        if first is None:
            first = (out.clone(), raw.clone())
        else:
            assert torch.equal(out, first[0]) and torch.equal(raw, first[1])


@pytest.mark.parametrize("variant", ["rowwin", "fast", "glds"])
def test_conv_kernels_random_shapes_vs_fp64(dev, variant):
    """60 seeded random layer shapes per kernel (filters 1..7 x 1..8, asymmetric padding, strides, rows of 1..40 pixels,
    Cin incl. the 16-channel tails, every tile width) against an fp64 convolution: the corner cases nobody thought of
    (tools/conv_rowwin_fuzz.py runs more of them: 600 row-window configurations without a mismatch in round 2)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("conv_fuzz", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                                                "tools", "conv_rowwin_fuzz.py"))
    fuzz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fuzz)
    done, bad = fuzz.run(seed=11, count=60, variant=variant, dev=dev, verbose=False)
    assert done == 60 and not bad, bad[:3]


def test_conv_block_major_k_order_matches_fp64_and_every_tile_width(dev):
    """Round 4: the default kernel with K order (32-channel block, tap) for unpadded multi-tap layers (csrc/conv_split.hip
    CBT; the stride-2 3x3 layers of Mixed_6a / 7a, chosen by SplitConv when the variant is left to "auto"): against an fp64
    convolution of the values the split tensor holds (the tolerance of every other kernel), within rounding of the
    tap-major kernel, the same bits for every tile width and run after run.  Shapes: the trunk's four, stride 1, 5x5, 2x3,
    M tails, one image, three output segments."""
    import torch.nn.functional as F
    from tise_toolbox_amd.conv_split import SplitConv, merge, new_split, split
    g = torch.Generator(device="cpu").manual_seed(77)
    for (n, H, W, Cin, Cout, kh, kw, st) in [(5, 35, 35, 288, 384, 3, 3, 2), (7, 35, 35, 96, 96, 3, 3, 2), (9, 17, 17, 192, 320, 3, 3, 2),
                                             (3, 17, 17, 192, 192, 3, 3, 2), (2, 9, 11, 64, 80, 3, 3, 1), (1, 13, 8, 32, 64, 5, 5, 2),
                                             (4, 6, 7, 96, 144, 2, 3, 1)]:
        x = (torch.rand((n, H, W, Cin), generator=g) * 2.0).to(dev)
        w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        xs = split(x)
        auto = SplitConv(w, b, (st, st), (0, 0), dev, variant="fast", korder="block")
        assert auto.korder == "block"
        oh, ow = auto.out_hw(H, W)
        outs = {}
        for tn in (2, 3, 4, 5):
            c = SplitConv(w, b, (st, st), (0, 0), dev, tn=tn, variant="fast", korder="block")
            o = new_split(n, oh, ow, Cout, dev)
            c(xs, [(0, Cout, o, 0, 0)])
            o2 = torch.zeros_like(o)
            c(xs, [(0, Cout, o2, 0, 0)])
            assert torch.equal(o, o2), (tn, H, W)
            outs[tn] = o
        assert all(torch.equal(outs[2], outs[t]) for t in (3, 4, 5)), (H, W, Cin, Cout)
        tap = SplitConv(w, b, (st, st), (0, 0), dev, variant="fast", korder="tap")
        ot = new_split(n, oh, ow, Cout, dev)
        tap(xs, [(0, Cout, ot, 0, 0)])
        ref = torch.relu(F.conv2d(merge(xs).double().permute(0, 3, 1, 2), w.double(), b.double(), st)).permute(0, 2, 3, 1)
        scale = ref.abs().max().item()
        err = (merge(outs[4]).double() - ref).abs().max().item()
        err_tap = (merge(ot).double() - ref).abs().max().item()
        assert err <= 4e-6 * scale and err_tap <= 4e-6 * scale, (H, W, Cin, Cout, err, err_tap, scale)
        assert (merge(outs[4]) - merge(ot)).abs().max().item() <= 4e-6 * scale
    # segments: split slice | split slice | raw fp32
    w = (torch.randn((160, 64, 3, 3), generator=g) * 0.05).to(dev); b = torch.zeros(160, device=dev)
    xs = split((torch.rand((2, 9, 9, 64), generator=g)).to(dev))
    c = SplitConv(w, b, (2, 2), (0, 0), dev, variant="fast", korder="block")
    a0, a1 = new_split(2, 4, 4, 64, dev), new_split(2, 4, 4, 64, dev)
    raw = torch.empty((2, 4, 4, 32), dtype=torch.float32, device=dev)
    c(xs, [(0, 64, a0, 0, 0), (64, 128, a1, 0, 0), (128, 160, raw, 0, 1)])
    ref = F.conv2d(merge(xs).double().permute(0, 3, 1, 2), w.double(), None, 2).permute(0, 2, 3, 1)
    got = torch.cat([merge(a0).double(), merge(a1).double(), torch.relu(raw.double())], -1)
    assert (got - torch.relu(ref)).abs().max().item() <= 4e-6 * ref.abs().max().item()
    with pytest.raises(ValueError):
        SplitConv(w, b, (1, 1), (1, 1), dev, variant="fast", korder="block")            # padded: the tap-major order only


def test_conv_default_variant_on_very_short_rows(dev):
    """The default variant picks the row-window kernel for stride-1 layers with KW > 1; on rows so short that its window
    would not fit (OW = 4 here) the layer falls back to the default kernel instead of failing."""
    from tise_toolbox_amd.conv_split import SplitConv, merge, rowwin_fits, split
    g = torch.Generator(device="cpu").manual_seed(4)
    x = (torch.rand((40, 4, 4, 64), generator=g) * 2.0).to(dev)
    w = (torch.randn((96, 64, 3, 3), generator=g) * (2.0 / 576) ** 0.5).to(dev)
    b = (torch.randn(96, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (1, 1), (1, 1), dev)
    assert conv.variant == "rowwin" and not rowwin_fits(4, 3) and rowwin_fits(35, 3) and rowwin_fits(8, 3)
    out = torch.zeros((40, 4, 4, 2 * 96), dtype=torch.float16, device=dev)
    conv(split(x), [(0, 96, out, 0, 0)])
    ref = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, 1)).permute(0, 2, 3, 1)
    assert (merge(out).double() - ref).abs().max().item() <= 4e-6 * ref.abs().max().item()
    assert conv._fallback is not None and conv._fallback.variant == "fast"


def test_conv_split_tile_width_does_not_change_results(dev):
    """Every tile width (tn = 1..5) accumulates every output element over K in the same order with the same MFMA
    sequence: the outputs must be bit-identical -- the tile width is a pure performance choice (conv_split.pick_tn).
    Includes Cin % 32 == 16 (paired tails) and Cout tails."""
    from tise_toolbox_amd.conv_split import SplitConv, split
    g = torch.Generator(device="cpu").manual_seed(21)
    for (n, H, W, Cin, Cout, kh, kw, st, pad) in [(9, 17, 17, 192, 192, 1, 7, 1, (0, 3)), (5, 35, 35, 48, 64, 5, 5, 1, (2, 2)),
                                                  (33, 8, 8, 448, 384, 3, 3, 1, (1, 1)), (3, 19, 19, 80, 208, 3, 3, 1, (0, 0))]:
        x = (torch.rand((n, H, W, Cin), generator=g) * 2.0).to(dev)
        w = (torch.randn((Cout, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        xs = split(x)
        for variant, tns in (("fast", (1, 2, 3, 4, 5)), ("rowwin", (2, 3, 4))):
            if variant == "rowwin" and kw < 2:
                continue
            ref = None
            for tn in tns:
                conv = SplitConv(w, b, (st, st), pad, dev, tn=tn, variant=variant)
                oh, ow = conv.out_hw(H, W)
                out = torch.full((n, oh, ow, 2 * Cout), 5.0, dtype=torch.float16, device=dev)
                conv(xs, [(0, Cout, out, 0, 0)])
                if ref is None:
                    ref = out
                else:
                    assert torch.equal(out, ref), (variant, tn, Cin, Cout)


def test_conv_pooled_input_is_bit_identical_to_pool_then_conv(dev):
    """conv_poolin_kernel (max_pool2d(3, 2) taken inside the operand load of a 1x1 convolution; the trunk's two stem
    pools, inception.py:61-71) against the two-kernel path it replaces -- tise_maxpool3s2_split_nhwc, then the default
    kernel: same re-split pooled values, same K order and MFMA sequence => every output bit equal; and against an
    fp64 max-pool + convolution.  Shapes: the trunk's two (147^2 x 64 -> 80; 71^2 x 192 -> 208 into four segments
    incl. the raw fp32 pool-branch slice), odd sizes, an M tail, one image, exact-tie windows (constant regions)."""
    import torch.nn.functional as F
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    from tise_toolbox_amd.trunk import SplitTrunk
    g = torch.Generator(device="cpu").manual_seed(33)
    for (n, H, W, Cin, Cout, segs_spec) in [(3, 147, 147, 64, 80, None), (2, 71, 71, 192, 208, (64, 112, 176, 208)),
                                            (5, 9, 12, 32, 48, None), (1, 3, 3, 96, 256, None), (7, 20, 7, 64, 144, (16, 144))]:
        x = (torch.rand((n, H, W, Cin), generator=g) * 3.0)
        x[:, : H // 2, : W // 2, : Cin // 2] = 1.25                  # constant windows: ties between taps
        x = x.to(dev)
        w = (torch.randn((Cout, Cin, 1, 1), generator=g) * (2.0 / Cin) ** 0.5).to(dev)
        b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        xs = split(x)
        conv = SplitConv(w, b, (1, 1), (0, 0), dev, variant="fast")
        oh, ow = conv.pooled_out_hw(H, W)

        def run(pooled):
            outs, segs = [], []
            bounds = [0] + list(segs_spec or (Cout,))
            for i, (c0, c1) in enumerate(zip(bounds[:-1], bounds[1:])):
                raw = segs_spec is not None and i == len(bounds) - 2            # last segment raw fp32 (pool branch)
                t = (torch.full((n, oh, ow, c1 - c0), 7.0, dtype=torch.float32, device=dev) if raw else
                     torch.full((n, oh, ow, 2 * (c1 - c0)), 7.0, dtype=torch.float16, device=dev))
                outs.append(t)
                segs.append((c0, c1, t, 0, 1 if raw else 0))
            if pooled:
                conv(xs, segs, pooled_input=True)
            else:
                conv(SplitTrunk._maxpool_split(xs), segs)
            return outs
        fused, two = run(True), run(False)
        for a, c in zip(fused, two):
            assert torch.equal(a, c), (H, W, Cin, Cout)
        # fp64 reference on the values the split tensor really holds
        xm = merge(xs).double().permute(0, 3, 1, 2)
        ref = F.conv2d(F.max_pool2d(xm, 3, 2), w.double(), None).permute(0, 2, 3, 1)
        bounds = [0] + list(segs_spec or (Cout,))
        got, want = [], []
        for o, c0, c1 in zip(fused, bounds[:-1], bounds[1:]):
            if o.dtype == torch.float32:                                  # raw segment: conv only (bias + ReLU follow the avg-pool)
                got.append(o.double()); want.append(ref[..., c0:c1])
            else:
                got.append(merge(o).double()); want.append(torch.relu(ref[..., c0:c1] + b.double()[c0:c1]))
        got, want = torch.cat(got, -1), torch.cat(want, -1)
        assert (got - want).abs().max().item() <= 4e-6 * max(1.0, want.abs().max().item()), (H, W, Cin, Cout)


def test_conv_pooled_output_is_bit_identical_to_conv_then_pool(dev):
    """Round 4 (inception.py:63-65: Conv2d_2b -> MaxPool2d(3, 2)): the register-weights kernel with the
    max-pool in its EPILOGUE (conv_pipe.hip, POOL instance: per-column running maxima in LDS while the workgroup walks the
    conv rows, horizontal pass per tile) against the two-kernel path it replaces -- the same kernel writing the full
    result, then tise_maxpool3s2_split_nhwc: the pooled (hi, lo) pairs must be equal bit for bit; and against an fp64
    convolution + max-pool.  Shapes: the trunk's (149^2 zero-bordered -> 147^2 -> 73^2) at batch sizes that cut the
    workgroups' runs of pool rows inside and between images, the narrowest and widest grids the kernel takes, odd / even
    output sizes, one image, constant regions (exact ties between taps), a channel slice of a wider destination."""
    import torch.nn.functional as F
    from tise_toolbox_amd.conv_split import SplitConv, merge, new_split, pool_output_fits, split
    from tise_toolbox_amd.trunk import SplitTrunk
    g = torch.Generator(device="cpu").manual_seed(41)
    w = (torch.randn((64, 32, 3, 3), generator=g) * (2.0 / 288) ** 0.5).to(dev)
    b = (torch.randn(64, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (1, 1), (0, 0), dev, variant="pipe", pipe_cfg=34)
    for (n, H, W) in [(3, 149, 149), (1, 149, 149), (37, 149, 149), (2, 7, 128), (5, 12, 131), (3, 10, 152), (4, 5, 140), (9, 149, 150)]:
        assert pool_output_fits(W, W - 2)
        x = torch.rand((n, H, W, 32), generator=g) * 2.0
        x[:, : H // 2, : W // 3] = 0.75                                # constant windows: ties between taps and between rows
        x[:, :, 0] = 0; x[:, :, -1] = 0; x[:, 0] = 0; x[:, -1] = 0    # (the trunk's input has a zero border; any input is legal)
        xs = split(x.to(dev))
        oh, ow = H - 2, W - 2
        ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
        full = new_split(n, oh, ow, 64, dev)
        conv(xs, [(0, 64, full, 0, 0)])
        want = SplitTrunk._maxpool_split(full)
        got = torch.full((n, ph, pw, 128), 9.0, dtype=torch.float16, device=dev)
        assert conv(xs, [(0, 64, got, 0, 0)], pool_output=True) == (ph, pw)
        assert torch.equal(got, want), (n, H, W)
        again = torch.zeros_like(got)
        conv(xs, [(0, 64, again, 0, 0)], pool_output=True)
        assert torch.equal(again, got)                                  # run-to-run (a race in V / HV would show here)
        # into channels 32..95 of a 128-channel tensor: the rest untouched
        wide = torch.full((n, ph, pw, 256), 3.0, dtype=torch.float16, device=dev)
        conv(xs, [(0, 64, wide, 32, 0)], pool_output=True)
        mw = merge(wide)
        assert torch.equal(mw[..., 32:96], merge(want)) and bool((mw[..., :32] == 3.0 + 3.0 / 2048).all()) and bool((mw[..., 96:] == 3.0 + 3.0 / 2048).all())
        ref = F.max_pool2d(torch.relu(F.conv2d(merge(xs).double().permute(0, 3, 1, 2), w.double(), b.double())), 3, 2).permute(0, 2, 3, 1)
        err = (merge(got).double() - ref).abs().max().item()
        assert err <= 4e-6 * ref.abs().max().item(), (n, H, W, err)


def test_pool_split_between_producer_and_consumer_is_bit_identical(dev):
    """Round 4 (inception.py:69-71: Conv2d_4a -> MaxPool2d(3, 2) -> Mixed_5b's 1x1 convolutions): the row-window kernel takes
    the HORIZONTAL half of the pool in its epilogue (POOLH: tiles overlap by two pixels so that every window lies in one
    tile), the pooled-input kernel the three VERTICAL taps (VT).  Against conv -> tise_maxpool3s2_split_nhwc -> default 1x1
    kernel and against conv -> nine-tap pooled-input kernel: every bit of the consumer's output, and the half-pooled map
    itself against the column maxima of the stored result.  Shapes: the trunk's (73^2 x 80 -> 71^2 x 192 -> 35^2 x 208 in
    four segments incl. the raw pool-branch slice), odd / even widths, padded 3x3 and 1x3, M tails, one image, ties."""
    import torch.nn.functional as F
    from tise_toolbox_amd.conv_split import SplitConv, merge, new_split, rowwin_fits, split
    from tise_toolbox_amd.trunk import SplitTrunk
    g = torch.Generator(device="cpu").manual_seed(52)
    for (n, H, W, Cin, Cmid, Cout, kh, kw, pad, segs_spec) in [
            (3, 73, 73, 80, 192, 208, 3, 3, (0, 0), (64, 112, 176, 208)), (2, 12, 15, 32, 96, 64, 3, 3, (1, 1), None),
            (1, 9, 10, 48, 96, 48, 3, 3, (0, 0), None), (5, 7, 21, 64, 192, 144, 1, 3, (0, 1), (16, 144)), (7, 17, 17, 32, 96, 256, 3, 3, (1, 1), None)]:
        x = torch.rand((n, H, W, Cin), generator=g) * 2.0
        x[:, : H // 2, : W // 2] = 0.5                                  # constant regions: ties between columns and rows
        xs = split(x.to(dev))
        w1 = (torch.randn((Cmid, Cin, kh, kw), generator=g) * (2.0 / (Cin * kh * kw)) ** 0.5).to(dev)
        b1 = (torch.randn(Cmid, generator=g) * 0.2).to(dev)
        w2 = (torch.randn((Cout, Cmid, 1, 1), generator=g) * (2.0 / Cmid) ** 0.5).to(dev)
        b2 = (torch.randn(Cout, generator=g) * 0.2).to(dev)
        prod = SplitConv(w1, b1, (1, 1), pad, dev, tn=3, variant="rowwin")
        cons = SplitConv(w2, b2, (1, 1), (0, 0), dev, variant="fast")
        oh, ow = prod.out_hw(H, W)
        assert rowwin_fits(ow, kw) and ow >= 3 and oh >= 3
        ph, pw = (oh - 3) // 2 + 1, (ow - 3) // 2 + 1
        full = new_split(n, oh, ow, Cmid, dev)
        prod(xs, [(0, Cmid, full, 0, 0)])
        half = torch.full((n, oh, pw, 2 * Cmid), 5.0, dtype=torch.float16, device=dev)
        assert prod(xs, [(0, Cmid, half, 0, 0)], pool_h=True) == (oh, pw)
        mf = merge(full)
        want_half = torch.stack([mf[:, :, 2 * j:2 * j + 3].amax(2) for j in range(pw)], 2)
        assert torch.equal(merge(half), want_half), (n, H, W)

        def run(mode):
            outs, segs = [], []
            bounds = [0] + list(segs_spec or (Cout,))
            for i, (c0, c1) in enumerate(zip(bounds[:-1], bounds[1:])):
                raw = segs_spec is not None and i == len(bounds) - 2
                t = (torch.full((n, ph, pw, c1 - c0), 7.0, dtype=torch.float32, device=dev) if raw else
                     torch.full((n, ph, pw, 2 * (c1 - c0)), 7.0, dtype=torch.float16, device=dev))
                outs.append(t)
                segs.append((c0, c1, t, 0, 1 if raw else 0))
            if mode == "split":
                assert cons(half, segs, pooled_input="v") == (ph, pw)
            elif mode == "nine":
                cons(full, segs, pooled_input=True)
            else:
                cons(SplitTrunk._maxpool_split(full), segs)
            return outs
        a, b, c = run("split"), run("nine"), run("pool")
        for u, v, t in zip(a, b, c):
            assert torch.equal(u, t) and torch.equal(v, t), (n, H, W, Cin, Cmid, Cout)


def test_split_trunk_fused_pools_do_not_change_a_bit(dev, monkeypatch):
    """The whole trunk with the stem max-pools fused into their consumers == the trunk with separate pool kernels."""
    from tise_toolbox_amd.inception import InceptionV3
    from tise_toolbox_amd.trunk import SplitTrunk
    m = InceptionV3([3], seed=0).to(dev).eval()
    x = torch.rand((5, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    fused = SplitTrunk(m, dev)
    assert fused.fuse_pool and fused.pool_in_2b and fused.pool2_split       # round 4: pool 1 in Conv2d_2b's epilogue, pool 2 split
    monkeypatch.setenv("TISE_POOL_PRODUCER", "0")
    monkeypatch.setenv("TISE_POOL2_SPLIT", "0")
    r3 = SplitTrunk(m, dev)                                                  # round 3's form: both pools in their consumers
    assert r3.fuse_pool and not r3.pool_in_2b and not r3.pool2_split
    monkeypatch.setenv("TISE_POOL_FUSE", "0")
    plain = SplitTrunk(m, dev)
    assert not plain.fuse_pool and not plain.pool_in_2b and not plain.pool2_split
    want = plain(x)
    assert torch.equal(fused(x), want) and torch.equal(r3(x), want)


def test_stem_mfma_kernel_matches_fp64_and_the_fma_kernel(dev):
    """The stem layer on the matrix cores (stem_mfma_u8_kernel: K = 27 taps as ONE split-precision K-step, weights in
    registers, the byte -> (hi, lo) table as one LDS gather per value) against an fp64 convolution of the table values and
    against the fp32-FMA kernel: odd sizes, the last pixels of the tensor (byte-wise run loader), M tails, one image."""
    import ctypes
    from tise_toolbox_amd import _lib
    from tise_toolbox_amd.conv_split import merge
    from tise_toolbox_amd.trunk import pack_stem_mfma
    g = torch.Generator(device="cpu").manual_seed(5)
    st = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    P = lambda t: ctypes.c_void_p(t.data_ptr())
    for (n, H, W) in [(3, 299, 299), (1, 3, 5), (2, 9, 7), (5, 31, 64), (1, 300, 299)]:
        u8 = torch.randint(0, 256, (n, H, W, 3), generator=g, dtype=torch.uint8).to(dev)
        lut = (torch.rand(3 * 256, generator=g) * 2.4 - 1.2).to(dev)
        w = (torch.randn((32, 3, 3, 3), generator=g) * (2.0 / 27) ** 0.5 * torch.exp2(torch.randint(-3, 4, (32, 1, 1, 1), generator=g).float())).to(dev)
        b = (torch.randn(32, generator=g) * 0.2).to(dev)
        oh, ow = (H - 3) // 2 + 1, (W - 3) // 2 + 1
        wsp, scale = pack_stem_mfma(w, dev)
        out = torch.full((n, oh, ow, 64), 3.0, dtype=torch.float16, device=dev)
        _lib.call("tise_stem_conv3x3s2_split_u8_mfma", P(u8), P(lut), n, H, W, P(wsp), P(scale), P(b), P(out), st())
        wf = w.permute(2, 3, 1, 0).contiguous()
        out_f = torch.full_like(out, 3.0)
        _lib.call("tise_stem_conv3x3s2_split_u8", P(u8), P(lut), n, H, W, P(wf), P(b), P(out_f), st())
        x = torch.stack([lut.view(3, 256)[c][u8[..., c].long()] for c in range(3)], 1).double()         # (n, 3, H, W)
        ref = torch.relu(torch.conv2d(x, w.double(), b.double(), 2)).permute(0, 2, 3, 1)
        scale_ref = ref.abs().max().item()
        assert (merge(out).double() - ref).abs().max().item() <= 4e-6 * scale_ref, (n, H, W)
        assert (merge(out_f).double() - ref).abs().max().item() <= 4e-6 * scale_ref


def test_split_trunk_batch_sizes_and_determinism(dev):
    """pool3 features must not depend on how images are batched, and must repeat bit for bit."""
    from tise_toolbox_amd.inception import InceptionV3
    from tise_toolbox_amd.trunk import SplitTrunk
    m = InceptionV3([3], seed=0).to(dev).eval()
    trunk = SplitTrunk(m, dev)
    x = torch.rand((11, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    full = trunk(x).clone()
    again = trunk(x)
    assert torch.equal(full, again)
    one = torch.cat([trunk(x[i:i + 1].contiguous(memory_format=torch.channels_last)) for i in range(11)], 0)
    chunks = torch.cat([trunk(x[:4].contiguous(memory_format=torch.channels_last)),
                        trunk(x[4:].contiguous(memory_format=torch.channels_last))], 0)
    assert torch.equal(one, full) and torch.equal(chunks, full)


def test_conv_split_rejects_misaligned_segments(dev):
    """Segments must tile [0, Cout) in order, start on multiples of 8 couts and land on 16-byte boundaries."""
    from tise_toolbox_amd import _lib
    from tise_toolbox_amd.conv_split import SplitConv, split
    w = torch.randn((64, 32, 1, 1), device=dev)
    conv = SplitConv(w, torch.zeros(64, device=dev), (1, 1), (0, 0), dev)
    x = split(torch.rand((2, 5, 5, 32), device=dev))
    out = torch.zeros((2, 5, 5, 2 * 80), dtype=torch.float16, device=dev)
    conv(x, [(0, 32, out, 0, 0), (32, 64, out, 40, 0)])                     # fine
    for segs in ([(0, 28, out, 0, 0), (28, 64, out, 32, 0)],               # boundary not a multiple of 8
                 [(0, 32, out, 4, 0), (32, 64, out, 40, 0)],               # destination offset not 16-byte aligned
                 [(0, 32, out, 0, 0), (40, 64, out, 40, 0)],               # gap
                 [(8, 64, out, 0, 0)]):                                     # does not start at 0
        with pytest.raises(_lib.TiseStatusError):
            conv(x, segs)


@pytest.mark.parametrize("case", [(149, 149, 32, (0, 0), 3), (35, 35, 32, (1, 1), 5), (23, 23, 32, (0, 0), 3), (9, 11, 64, (1, 1), 2),
                                  (149, 149, 32, (0, 0), 40), (147, 147, 64, (1, 1), 9)])
@pytest.mark.parametrize("cfg", [34])
def test_conv_win32_sliding_window_kernel(dev, case, cfg):
    """conv_pipe.hip configuration 34 (REGISTER-resident weights, input through a sliding LDS ring, all eight waves compute
    and finish their own tiles, 32 or 64 couts in one launch) for the 32-channel 3x3 stride-1 layers.  Against fp64 (valid
    and padded borders, image boundaries inside tiles, one tile per workgroup up to 27, tails of the grid, odd tile counts,
    64 couts, three destination segments), bit-identical over repeated runs (ring reuse / barriers) and bit-identical to
    the generic kernel (same K order and MFMA sequence)."""
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    H, W, Cout, pad, n = case
    g = torch.Generator(device="cpu").manual_seed(H + Cout + n)
    x = (torch.rand((n, H, W, 32), generator=g) * 3.0).to(dev)
    w = (torch.randn((Cout, 32, 3, 3), generator=g) * (2.0 / 288) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * 0.2).to(dev)
    conv = SplitConv(w, b, (1, 1), pad, dev, variant="pipe", pipe_cfg=cfg)
    generic = SplitConv(w, b, (1, 1), pad, dev, variant="glds")
    oh, ow = conv.out_hw(H, W)
    ref_lin = torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), None, 1, pad).permute(0, 2, 3, 1)
    ref = torch.relu(ref_lin + b.double())
    scale = ref.abs().max().item()
    xs = split(x)
    first = None
    for rep in range(4):
        out = torch.zeros((n, oh, ow, 2 * (Cout + 32)), dtype=torch.float16, device=dev)
        raw = torch.zeros((n, oh, ow, 16), dtype=torch.float32, device=dev)
        segs = [(0, 16, out, 16, 0), (16, 32, raw, 0, 1)] + ([(32, Cout, out, 64, 0)] if Cout > 32 else [])
        conv(xs, segs)
        got = merge(out)
        assert (got[..., 16:32].double() - ref[..., 0:16]).abs().max().item() <= 4e-6 * scale
        if Cout > 32:
            assert (got[..., 64:].double() - ref[..., 32:]).abs().max().item() <= 4e-6 * scale
        assert (raw.double() - ref_lin[..., 16:32]).abs().max().item() <= 4e-6 * scale
        assert got[..., :16].abs().max().item() == 0 and got[..., 32:64].abs().max().item() == 0
        if first is None:
            first = (out.clone(), raw.clone())
            out_g, raw_g = torch.zeros_like(out), torch.zeros_like(raw)
            generic(xs, [(a, c, out_g if m == 0 else raw_g, off, m) for (a, c, _, off, m) in segs])
            assert torch.equal(out, out_g) and torch.equal(raw, raw_g)
        else:
            assert torch.equal(out, first[0]) and torch.equal(raw, first[1])


@pytest.mark.parametrize("cfg", [34])
def test_conv_sliding_window_writes_into_a_zero_bordered_buffer(dev, cfg, monkeypatch):
    """args->out_hp (round 3): Conv2d_2a writes its result into the interior of a zero-bordered buffer and the padded
    Conv2d_2b runs as a VALID convolution over it (no tap masks).  (i) the interior equals the plain output bit for
    bit and the border is untouched; (ii) valid conv over the bordered buffer == padded conv over the plain tensor,
    bit for bit (the masked taps contributed exact zeros); (iii) the trunk with and without the buffer agrees bit for
    bit; (iv) kernels without a grid epilogue refuse the offset form."""
    from tise_toolbox_amd import _lib
    from tise_toolbox_amd.conv_split import SplitConv, split
    from tise_toolbox_amd.inception import InceptionV3
    from tise_toolbox_amd.trunk import SplitTrunk
    g = torch.Generator(device="cpu").manual_seed(77)
    for (n, H, W) in [(3, 29, 31), (2, 149, 149)]:
        x = (torch.rand((n, H, W, 32), generator=g) * 3.0).to(dev)
        w1 = (torch.randn((32, 32, 3, 3), generator=g) * (2.0 / 288) ** 0.5).to(dev)
        w2 = (torch.randn((64, 32, 3, 3), generator=g) * (2.0 / 288) ** 0.5).to(dev)
        b1, b2 = (torch.randn(32, generator=g) * 0.2).to(dev), (torch.randn(64, generator=g) * 0.2).to(dev)
        xs = split(x)
        c1 = SplitConv(w1, b1, (1, 1), (0, 0), dev, variant="pipe", pipe_cfg=34)
        oh, ow = c1.out_hw(H, W)
        plain = torch.empty((n, oh, ow, 64), dtype=torch.float16, device=dev)
        c1(xs, [(0, 32, plain, 0, 0)])
        buf = torch.full((n, oh + 2, ow + 2, 64), 9.0, dtype=torch.float16, device=dev)
        c1(xs, [(0, 32, buf, 0, 0)], out_pad=(oh + 2, ow + 2, 1, 1))
        assert torch.equal(buf[:, 1:-1, 1:-1], plain)
        border = buf.clone(); border[:, 1:-1, 1:-1] = 9.0
        assert (border == 9.0).all()
        buf[:, 0] = 0; buf[:, -1] = 0; buf[:, :, 0] = 0; buf[:, :, -1] = 0
        padded = SplitConv(w2, b2, (1, 1), (1, 1), dev, variant="pipe", pipe_cfg=34)
        valid = SplitConv(w2, b2, (1, 1), (0, 0), dev, variant="pipe", pipe_cfg=34)
        o1 = torch.empty((n, oh, ow, 128), dtype=torch.float16, device=dev)
        o2 = torch.empty_like(o1)
        padded(plain, [(0, 64, o1, 0, 0)])
        valid(buf, [(0, 64, o2, 0, 0)])
        assert torch.equal(o1, o2)
        fast = SplitConv(w1, b1, (1, 1), (0, 0), dev, variant="fast")
        with pytest.raises(_lib.TiseStatusError):
            _raw_out_pad_call(fast, xs, buf, oh, ow)
    m = InceptionV3([3], seed=0).to(dev).eval()
    x = torch.rand((3, 3, 299, 299), device=dev).contiguous(memory_format=torch.channels_last)
    with_buf = SplitTrunk(m, dev)
    assert with_buf.pad2b
    os.environ["TISE_CONV_PADBUF"] = "0"
    try:
        without = SplitTrunk(m, dev)
    finally:
        del os.environ["TISE_CONV_PADBUF"]
    assert not without.pad2b
    a, b = with_buf(x).clone(), without(x).clone()
    assert torch.equal(a, b) and torch.equal(with_buf(x), a)            # second call reuses the persistent buffer


def _raw_out_pad_call(conv, xs, buf, oh, ow):
    """SplitConv refuses out_pad for non-sliding-window variants in Python; go through the C-ABI to see the library refuse it."""
    import ctypes
    from tise_toolbox_amd import _lib
    from tise_toolbox_amd.conv_split import ConvArgs
    n, h, w, _ = xs.shape
    a = ConvArgs()
    a.x = xs.data_ptr(); a.w = conv.w_fast.data_ptr(); a.scale = conv.scale.data_ptr(); a.bias = conv.bias.data_ptr()
    a.N, a.H, a.W, a.Cin, a.KH, a.KW, a.SH, a.SW, a.PH, a.PW, a.OH, a.OW = n, h, w, 32, 3, 3, 1, 1, 0, 0, oh, ow
    a.Cout, a.K, a.Kpad, a.M, a.nseg = conv.cout, conv.k, conv.kpad, n * oh * ow, 1
    a.seg[0].c0, a.seg[0].c1, a.seg[0].dst, a.seg[0].ld, a.seg[0].off, a.seg[0].mode = 0, conv.cout, buf.data_ptr(), 32, 0, 0
    a.out_hp, a.out_wp, a.out_y0, a.out_x0 = oh + 2, ow + 2, 1, 1
    _lib.call("tise_conv_split_f16", ctypes.byref(a), conv.tn | 128, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))


# ------------------------------------------------------------------------------------------- split format: range
@pytest.mark.parametrize("case", [(17, 17, 192, 224, 1, 7, (0, 3)), (35, 35, 64, 96, 3, 3, (1, 1)), (8, 8, 448, 384, 3, 3, (1, 1)),
                                  (73, 73, 80, 192, 3, 3, (0, 0)), (17, 17, 768, 704, 1, 1, (0, 0))])
def test_conv_split_dynamic_range_and_heavy_tails_vs_fp64(dev, case):
    """Real checkpoints do not have the calibrated stand-in magnitudes: activations spanning 1e-6 .. 3e4 (log-uniform,
    i.e. every binade of the fp16 range and its subnormals is populated) and heavy-tailed weights (Student t, 2 dof:
    single weights 100x the typical one, per-channel scales spread over 4 decades).  Against an fp64 convolution the
    split kernel must stay at the error of an exact-fp32 convolution (MIOpen's) or better, per OUTPUT CHANNEL (each
    channel has its own scale), and the range guard must stay silent."""
    from tise_toolbox_amd import device
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    H, W, Cin, Cout, kh, kw, pad = case
    g = torch.Generator(device="cpu").manual_seed(H + Cin + Cout)
    n = 9
    x = torch.exp(torch.empty((n, H, W, Cin)).uniform_(float(np.log(1e-6)), float(np.log(3e4)), generator=g))
    x = (x * (torch.rand((n, H, W, Cin), generator=g) > 0.3)).to(dev)                    # ReLU-like zeros
    t = torch.distributions.StudentT(2.0).sample((Cout, Cin, kh, kw))
    chan = torch.exp(torch.empty(Cout).uniform_(float(np.log(1e-4)), 0.0, generator=g))    # per-channel scale, 4 decades
    w = (t * chan.view(-1, 1, 1, 1) / (Cin * kh * kw) ** 0.5).to(dev)
    b = (torch.randn(Cout, generator=g) * chan).to(dev)
    # the OUTPUT is the next layer's activation: bring its maximum to 3e4 as well (inside the fp16 range)
    peak = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, pad)).max().item()
    w, b = w * (3e4 / peak), b * (3e4 / peak)
    device.read_split_overflow()
    conv = SplitConv(w, b, (1, 1), pad, dev)
    oh, ow = conv.out_hw(H, W)
    out = torch.zeros((n, oh, ow, 2 * Cout), dtype=torch.float16, device=dev)
    conv(split(x), [(0, Cout, out, 0, 0)])
    assert not device.read_split_overflow()
    got = merge(out).double()
    ref = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2).double(), w.double(), b.double(), 1, pad)).permute(0, 2, 3, 1)
    f32 = torch.relu(torch.conv2d(x.permute(0, 3, 1, 2), w, b, 1, pad)).permute(0, 2, 3, 1).double()
    scale = ref.abs().amax(dim=(0, 1, 2)).clamp_min(1e-300)                                # per output channel
    e_split = ((got - ref).abs().amax(dim=(0, 1, 2)) / scale).max().item()
    e_f32 = ((f32 - ref).abs().amax(dim=(0, 1, 2)) / scale).max().item()
    print(f"{case}: split {e_split:.2e}  fp32 conv {e_f32:.2e}")
    assert torch.isfinite(got).all()
    assert e_split <= max(4e-6, 2.0 * e_f32), (e_split, e_f32)


def test_split_format_small_value_floor(dev):
    """What the format does below the fp16 normal range: hi becomes subnormal, lo keeps 11 bits of the residual, so the
    ABSOLUTE error floor is ~1.5e-11 -- fp32-relative accuracy down to |v| ~ 2.5e-4, degrading gracefully below."""
    from tise_toolbox_amd.conv_split import merge, split
    for s, tol in ((1.0, 3e-7), (1e-3, 3e-7), (1e-5, 3e-6), (1e-7, 3e-4)):
        v = (torch.rand(100000, device=dev) + 0.5) * s
        rel = ((merge(split(v)).double() - v.double()).abs() / v.double()).max().item()
        assert rel <= tol, (s, rel)
        assert ((merge(split(v)).double() - v.double()).abs()).max().item() <= max(3e-7 * s, 2e-11)


def test_split_overflow_guard_fires_and_clears(dev):
    """A value above 65504 (or a NaN) converted into a split tensor raises the device flag: conv epilogue, stem, and the
    engine turns it into FloatingPointError instead of silently carrying +inf through the trunk."""
    from tise_toolbox_amd import device
    from tise_toolbox_amd.conv_split import SplitConv, merge, split
    device.read_split_overflow()
    x = torch.full((2, 9, 9, 32), 300.0, device=dev)
    w = torch.full((64, 32, 1, 1), 1.0, device=dev)
    b = torch.zeros(64, device=dev)
    conv = SplitConv(w, b, (1, 1), (0, 0), dev)
    out = torch.zeros((2, 9, 9, 2 * 64), dtype=torch.float16, device=dev)
    conv(split(x), [(0, 64, out, 0, 0)])                       # 32 * 300 = 9 600: fine
    assert not device.read_split_overflow() and float(merge(out).max()) == 9600.0
    conv(split(x * 8), [(0, 64, out, 0, 0)])                   # 76 800 > 65 504
    assert not torch.isfinite(merge(out)).all()                # hi = +inf, lo = -inf: the halves merge to NaN
    assert device.read_split_overflow()
    assert not device.read_split_overflow()                    # read-and-clear
    with pytest.raises(FloatingPointError, match="fp16 range"):
        conv(split(x * 8), [(0, 64, out, 0, 0)])
        device.check_split_overflow()
    c2 = SplitConv(w, b, (1, 1), (0, 0), dev, variant="glds")
    c2(split(x * 8), [(0, 64, out, 0, 0)])
    assert device.read_split_overflow(), "glds"
    c3 = SplitConv(torch.full((32, 32, 3, 3), 1.0, device=dev), torch.zeros(32, device=dev), (1, 1), (1, 1), dev, variant="pipe", pipe_cfg=34)
    out3 = torch.zeros((2, 9, 9, 2 * 32), dtype=torch.float16, device=dev)
    c3(split(x * 8), [(0, 32, out3, 0, 0)])
    assert device.read_split_overflow(), "regw32"
    # engine level: huge stand-in scale in the first conv -> FloatingPointError at statistics() time
    from tise_toolbox_amd.engine import RealismEngine
    from tise_toolbox_amd.inception import InceptionV3
    m = InceptionV3([3], seed=0)
    with torch.no_grad():
        m.blocks[0][0].bn.weight.mul_(1e6)
    eng = RealismEngine(dims=2048, model=m)
    eng.begin()
    eng.step_u8(torch.randint(0, 256, (4, 64, 64, 3), dtype=torch.uint8, device=dev))
    with pytest.raises(FloatingPointError):
        eng.statistics()


def test_split_trunk_with_uncalibrated_checkpoint_matches_fp32_trunk(dev, tmp_path, monkeypatch):
    """--weights path through the HIP trunk with magnitudes the stand-ins never have: every BatchNorm gamma / beta of
    the seeded net perturbed by log-normal factors (sigma 0.7: per-channel scales from 0.2x to 5x), saved as a
    torchvision-format state_dict, loaded strictly, and run through the split-fp16 trunk and the exact-fp32 (MIOpen)
    trunk: features agree to 1e-4 of their scale, the range guard stays silent."""
    from tise_toolbox_amd import device
    from tise_toolbox_amd.engine import RealismEngine
    from tise_toolbox_amd.inception import build_inception3
    net = build_inception3(seed=0)
    g = torch.Generator().manual_seed(3)
    sd = net.state_dict()
    for k in sd:
        if k.endswith("bn.weight") or k.endswith("bn.bias"):
            # exp(s z - s^2): E[f^2] = 1, so the layer-to-layer signal power is preserved on average while single
            # channels run 0.1x .. 4x their calibrated scale
            sd[k] = sd[k] * torch.exp(0.7 * torch.randn(sd[k].shape, generator=g) - 0.49)
    path = tmp_path / "perturbed_inception.pth"
    torch.save(sd, path)
    imgs = torch.from_numpy(_cases.smooth_images(12, 256, 256, seed=8)).to(dev)
    monkeypatch.setenv("TISE_CONV", "split")
    a = RealismEngine(dims=2048, weights=str(path), with_logits=True)
    monkeypatch.setenv("TISE_CONV", "miopen")
    monkeypatch.setenv("TISE_MIOPEN_FIND", "0")
    b = RealismEngine(dims=2048, weights=str(path), with_logits=True)
    device.read_split_overflow()
    fa, la = a.features_from_u8(imgs)
    fb, lb = b.features_from_u8(imgs)
    a.check_numerics()
    scale = fb.abs().max().item()
    err = (fa - fb).abs().max().item()
    print("perturbed checkpoint: feature scale", scale, "max err", err)
    assert np.isfinite(scale) and scale > 0 and err <= 1e-4 * scale
    assert (la - lb).abs().max().item() <= 1e-3 * max(1.0, lb.abs().max().item())
    # the same checkpoint through the CPU ORACLE on 8 of the images (PIL-exact resize, unfolded BatchNorm, CPU fp32)
    from oracle import inception_oracle, resize_oracle
    sd_cpu = {k: v.float() for k, v in torch.load(path).items()}
    xin = torch.from_numpy(np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299))
                                     for im in imgs[:8].cpu().numpy()]))
    o = inception_oracle.inception_forward(sd_cpu, xin)[3].flatten(1)
    err_o = (fa[:8].cpu() - o).abs().max().item()
    print("perturbed checkpoint: HIP trunk vs CPU oracle max err", err_o, "of scale", o.abs().max().item())
    assert err_o <= 2e-4 * o.abs().max().item()
    lo = inception_oracle.logits_from_pool3(sd_cpu, inception_oracle.inception_forward(sd_cpu, xin)[3], bias=False)   # coco head, :104-105
    assert (la[:8].cpu() - lo).abs().max().item() <= 1e-3 * max(1.0, lo.abs().max().item())


def test_classifier_layer_runs_as_a_split_convolution(cuda_device, monkeypatch):
    """Round 5: the IS* logits = pool3 x W (inception_score_star_coco.py:104-105) without a library GEMM: the global-mean
    kernel also writes pool3 as a split row (tise_split_mean_both_nhwc, bit-identical fp32 output, exact split of it) and the
    classifier layer runs as a 1x1 split-precision convolution with raw fp32 out (SplitTrunk.fc_logits).  Against an fp64
    product of the same rows: <= 2e-6 of the logit scale (the torch / hipBLASLt fp32 GEMM it replaces: same class); the bias of
    the bird / ois rules is added afterwards; TISE_FC=torch keeps the library GEMM and the two engines agree."""
    import ctypes
    from tise_toolbox_amd import _lib, conv_split as cs
    from tise_toolbox_amd.engine import RealismEngine
    dev = cuda_device
    g = torch.Generator(device="cpu").manual_seed(5)
    a = cs.split((torch.rand((37, 8, 8, 2048), generator=g) * torch.rand((1, 1, 1, 2048), generator=g) * 3).to(dev))
    st = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    f0 = torch.empty((37, 2048), dtype=torch.float32, device=dev)
    f1 = torch.empty_like(f0)
    fs = torch.empty((37, 1, 1, 4096), dtype=torch.float16, device=dev)
    _lib.call("tise_split_mean_nhwc", ctypes.c_void_p(a.data_ptr()), 37, 64, 2048, ctypes.c_void_p(f0.data_ptr()), st)
    _lib.call("tise_split_mean_both_nhwc", ctypes.c_void_p(a.data_ptr()), 37, 64, 2048, ctypes.c_void_p(f1.data_ptr()),
              ctypes.c_void_p(fs.data_ptr()), st)
    assert torch.equal(f0, f1)
    assert torch.equal(fs.view(37, 4096), cs.split(f1))                  # the split row is exactly split(fp32 mean)
    imgs = torch.from_numpy(_cases.smooth_images(24, 256, 256, seed=12)).to(dev)
    eng = RealismEngine(dims=2048, seed=0, with_logits=True)
    assert eng.fused.sfc is not None
    feats, logits = eng.features_from_u8(imgs)                           # coco rule: no bias
    w = eng.model.fc.weight.detach().double()
    want = feats.double() @ w.t()
    scale = float(want.abs().max())
    err = float((logits.double() - want).abs().max()) / scale
    lib = torch.nn.functional.linear(feats, eng.model.fc.weight)
    err_lib = float((lib.double() - want).abs().max()) / scale
    print(f"classifier layer as split convolution: error {err:.2e} of the logit scale (library fp32 GEMM {err_lib:.2e})")
    assert err <= 2e-6
    eng.begin(n_total=24, rule="bird", temperature=0.5980541706085205)
    lb = eng.features_from_u8(imgs)[1]
    assert float((lb.double() - (want + eng.model.fc.bias.double())).abs().max()) / scale <= 2e-6
    monkeypatch.setenv("TISE_FC", "torch")
    ref = RealismEngine(dims=2048, seed=0, with_logits=True)
    assert ref.fused.sfc is None
    f2, l2 = ref.features_from_u8(imgs)
    assert torch.equal(f2, feats) and float((l2 - logits).abs().max()) / scale <= 3e-6
