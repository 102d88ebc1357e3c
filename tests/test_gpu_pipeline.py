"""GPU end-to-end parity: uint8 images -> resize -> InceptionV3 -> statistics -> FID / IS* on the
device, against the CPU oracle run on the SAME images and the SAME (seeded) weights.

Tolerances are north_star's: |dFID| <= 1e-3, |dIS| <= 1e-4.  The conv stack itself is third-party
arithmetic (torchvision) with no reference golden vectors: "parity unpinned" for that stage means
GPU-vs-own-CPU-fp32, which is what is checked here.
"""
import os

import numpy as np
import pytest
import torch

from oracle import fid_oracle, inception_oracle, is_oracle, resize_oracle
from tests import _cases
from tise_toolbox_amd.weights import SYNTHETIC_TAG

pytestmark = pytest.mark.gpu

N_GEN, N_REF = 48, 40


@pytest.fixture(scope="module")
def setup(cuda_device):
    from tise_toolbox_amd.engine import RealismEngine
    from tise_toolbox_amd.inception import build_inception3
    eng = RealismEngine(dims=2048, seed=0, with_logits=True)
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    gen = _cases.smooth_images(N_GEN, 256, 256, seed=0)
    ref = _cases.smooth_images(N_REF, 256, 256, seed=1, shift=0.15)

    torch.set_num_threads(min(16, torch.get_num_threads()))    # CPU oracle forward: 3x faster at 16 threads than at 128

    def oracle_feats(imgs):
        xs = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299)) for im in imgs])
        feats, logits, logits_b = [], [], []
        for i in range(0, len(xs), 8):
            o = inception_oracle.inception_forward(sd, torch.from_numpy(xs[i:i + 8]))[3]
            feats.append(o.flatten(1).numpy())
            # the IS* head of inception_score_star_coco.py:104-105: pool3 x W, the graph's bias is NOT added
            logits.append(inception_oracle.logits_from_pool3(sd, o, bias=False).numpy())
            logits_b.append(inception_oracle.logits_from_pool3(sd, o, bias=True).numpy())
        return np.concatenate(feats), np.concatenate(logits), np.concatenate(logits_b)

    fg, lg, lg_bias = oracle_feats(gen)
    fr, _, _ = oracle_feats(ref)
    return dict(eng=eng, gen=gen, ref=ref, fg=fg, lg=lg, lg_bias=lg_bias, fr=fr, dev=cuda_device, sd=sd)


def test_features_match_cpu_fp32(setup):
    eng, dev = setup["eng"], setup["dev"]
    feats, logits = eng.features_from_u8(torch.as_tensor(setup["gen"], device=dev))
    f = feats.cpu().numpy()
    err = np.abs(f - setup["fg"]).max()
    scale = np.abs(setup["fg"]).max()
    print("pool3 max abs err", err, "scale", scale)
    assert err <= 2e-4 * scale            # fp32 conv stack, different summation order + BN folding
    lerr = np.abs(logits.cpu().numpy() - setup["lg"]).max()
    assert lerr <= 2e-3 * max(1.0, np.abs(setup["lg"]).max())


def test_fid_end_to_end_matches_oracle(setup):
    """Device: accumulate -> finalize -> Frechet.  Oracle: np.mean/np.cov -> scipy sqrtm (reference form).
    N << d: both covariances are rank deficient (BASELINE config 1 situation)."""
    from tise_toolbox_amd import fid_score
    eng, dev = setup["eng"], setup["dev"]
    eng.begin(n_total=N_GEN)
    for i in range(0, N_GEN, 16):
        eng.step_u8(torch.as_tensor(setup["gen"][i:i + 16], device=dev), i)
    eng.reduce()
    mu_g, sig_g = eng.statistics()
    is_dev = eng.inception_score()
    eng.begin(n_total=N_REF)
    for i in range(0, N_REF, 8):
        eng.step_u8(torch.as_tensor(setup["ref"][i:i + 8], device=dev), i)
    mu_r, sig_r = eng.statistics()
    fid_dev = fid_score.calculate_frechet_distance(mu_g, sig_g, mu_r, sig_r)

    m1, s1 = fid_oracle.calculate_activation_statistics(setup["fg"])
    m2, s2 = fid_oracle.calculate_activation_statistics(setup["fr"])
    fid_cpu = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    print("FID device", fid_dev, "oracle", fid_cpu)
    assert abs(fid_dev - fid_cpu) <= 1e-3
    is_cpu = is_oracle.inception_score_from_logits(setup["lg"], is_oracle.T_COCO, 10, "coco", dtype=np.float32)
    print("IS device", is_dev, "oracle", is_cpu)
    assert abs(is_dev[0] - is_cpu[0]) <= 1e-4 and abs(is_dev[1] - is_cpu[1]) <= 1e-4


def test_reference_api_functions(setup, tmp_path, capsys):
    """The drop-in functions with the reference's calling conventions (float CHW batches in [0,1])."""
    from tise_toolbox_amd import fid_score
    from tise_toolbox_amd.inception import InceptionV3
    model = InceptionV3([3], seed=0)
    bs = 8
    x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299)) for im in setup["gen"][:20]])
    loader = [torch.from_numpy(x[i * bs:(i + 1) * bs]) for i in range(20 // bs)]       # drop_last: 16 of 20 used
    act = fid_score.get_activations(loader, model, bs, 2048, cuda=True, verbose=True)
    assert " done" in capsys.readouterr().out
    assert act.dtype == np.float64 and act.shape == (16, 2048)
    assert np.abs(act - setup["fg"][:16]).max() <= 2e-4 * np.abs(setup["fg"]).max()
    mu, sigma = fid_score.calculate_activation_statistics(loader, model, bs, 2048, cuda=True, verbose=False)
    mu_ref, sigma_ref = fid_oracle.calculate_activation_statistics(act)
    # two separate forward passes: MIOpen's atomic split-K kernels sum in a run-dependent order, so
    # features repeat only to fp32 rounding (amplified by the 94-layer random stack): 2e-5 absolute
    np.testing.assert_allclose(mu, mu_ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(sigma, sigma_ref, rtol=0, atol=2e-5)
    from tise_toolbox_amd import _lib
    with pytest.raises(_lib.TiseLibraryError):
        fid_score.get_activations(loader, model, bs, 2048, cuda=False)
    # lower blocks: spatial maps are average-pooled (fid_score.py:110-111)
    m192 = InceptionV3([1], seed=0)
    a192 = fid_score.get_activations(loader[:1], m192, bs, 192, cuda=True, verbose=False)
    assert a192.shape == (8, 192)


def test_cli_end_to_end(setup, tmp_path, capsys):
    """fid_score CLI on directories of PNGs + an .npz, result text and value vs the oracle pipeline."""
    from PIL import Image
    from tise_toolbox_amd import fid_score, img_data
    gdir, rdir = tmp_path / "gen", tmp_path / "ref" / "sub"
    gdir.mkdir()
    rdir.mkdir(parents=True)
    for i in range(21):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:05d}.png")
    (gdir / "notes.txt").write_text("ignored")
    for i in range(18):
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:05d}.png")
    npz = tmp_path / "ref_stats.npz"
    out1 = tmp_path / "o1.txt"
    v1 = fid_score.main(["--batch-size", "5", "--path1", str(tmp_path / "ref"), "--path2", str(gdir),
                         "--saved_file", str(out1), "--gpu", "0", "--num-workers", "2", "--save-stats", str(tmp_path / "gen_stats.npz"),
                         "--synthetic-weights"])
    assert out1.read_text() == f"FID: {v1}" + SYNTHETIC_TAG
    # without --weights / --synthetic-weights there is no silent stand-in (the reference always loads pretrained)
    with pytest.raises(RuntimeError, match="no parameters for InceptionV3"):
        fid_score.main(["--batch-size", "5", "--path1", str(tmp_path / "ref"), "--path2", str(gdir)])
    # oracle pipeline on the same files, same walk order, same drop-last rule
    def oracle_stats(root, bs):
        files = img_data.get_filenames(str(root))
        files = files[:fid_oracle.n_used_images(len(files), bs)]
        idx = []
        pool = {"gen": setup["gen"], "ref": setup["ref"]}
        feats = setup["fg"] if "gen" in str(root) else setup["fr"]
        for f in files:
            idx.append(int(os.path.basename(f).split(".")[0]))
        return fid_oracle.calculate_activation_statistics(feats[idx])
    m1, s1 = oracle_stats(tmp_path / "ref", 5)
    m2, s2 = oracle_stats(gdir, 5)
    want = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    assert abs(v1 - want) <= 1e-3
    # .npz branch (fid_score.py:200-203) with stats written by --save-stats
    np.savez(npz, mu=m1, sigma=s1)
    out2 = tmp_path / "o2.txt"
    v2 = fid_score.main(["--batch-size", "5", "--path1", str(npz), "--path2", str(gdir), "--saved_file", str(out2),
                         "--label", "O-FID", "--num-workers", "0", "--synthetic-weights"])
    assert out2.read_text().startswith("O-FID: ") and abs(v2 - want) <= 1e-3
    g = np.load(tmp_path / "gen_stats.npz")
    np.testing.assert_allclose(g["mu"], m2, atol=1e-5)
    with pytest.raises(RuntimeError, match="Invalid path"):
        fid_score.main(["--path1", str(tmp_path / "missing"), "--path2", str(gdir), "--synthetic-weights"])
    # STATS-ONLY mode (section 8 f1): --path1 omitted -> {mu, sigma} of --path2, no Frechet distance solved
    fid_score.calculate_frechet_distance.last_result = "untouched"
    so = tmp_path / "only_stats.npz"
    assert fid_score.main(["--batch-size", "5", "--path2", str(gdir), "--save-stats", str(so), "--num-workers", "0",
                           "--synthetic-weights"]) is None
    assert fid_score.calculate_frechet_distance.last_result == "untouched"
    g2 = np.load(so)
    assert g2["mu"].dtype == np.float64 and g2["sigma"].shape == (2048, 2048)
    # against np.mean / np.cov of the DEVICE features of the same files: accumulation parity at rounding level
    np.testing.assert_array_equal(g2["mu"], g["mu"])
    np.testing.assert_array_equal(g2["sigma"], g["sigma"])
    np.testing.assert_allclose(g2["sigma"], s2, rtol=0, atol=2e-5 * np.abs(s2).max())       # vs the CPU-oracle features
    files = img_data.get_filenames(str(gdir))
    files = files[:fid_oracle.n_used_images(len(files), 5)]
    from tise_toolbox_amd.engine import RealismEngine
    eng = RealismEngine(dims=2048, seed=0)
    idx = [int(os.path.basename(f).split(".")[0]) for f in files]
    fdev = torch.cat([eng.features_from_u8(torch.from_numpy(setup["gen"][idx[i:i + 5]]).to(eng.device))[0]
                      for i in range(0, len(idx), 5)]).double().cpu().numpy()          # same batches as the CLI run
    mu_d, sig_d = fid_oracle.calculate_activation_statistics(fdev)
    np.testing.assert_allclose(g2["mu"], mu_d, rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(g2["sigma"], sig_d, rtol=0, atol=1e-12 * np.abs(sig_d).max())
    # and the saved file feeds the .npz branch
    v3 = fid_score.main(["--batch-size", "5", "--path1", str(so), "--path2", str(gdir), "--num-workers", "0", "--synthetic-weights"])
    assert abs(v3) <= 1e-4                               # FID(X, X) on rank-20 covariances: rounding of the zero eigenvalues
    with pytest.raises(SystemExit):
        fid_score.main(["--path2", str(gdir), "--synthetic-weights"])                       # neither --path1 nor --save-stats


def test_is_cli(setup, tmp_path):
    from PIL import Image
    from tise_toolbox_amd import inception_score as isc, img_data
    d = tmp_path / "imgs"
    d.mkdir()
    for i in range(30):
        Image.fromarray(setup["gen"][i]).save(d / f"{i:05d}.png")
    out = tmp_path / "is.txt"
    mean, std = isc.main(["--image_folder", str(d), "--saved_file", str(out), "--batch-size", "7", "--synthetic-weights"])
    assert out.read_text() == "[Inception Score] mean: {:.5f} std: {:.5f}".format(mean, std) + SYNTHETIC_TAG
    files = img_data.get_filenames(str(d))
    idx = [int(os.path.basename(f).split(".")[0]) for f in files]
    want = is_oracle.inception_score_from_logits(setup["lg"][idx], is_oracle.T_COCO, 10, "coco", dtype=np.float32)
    assert abs(mean - want[0]) <= 1e-4 and abs(std - want[1]) <= 1e-4


def test_is_star_coco_head_has_no_bias(setup, tmp_path):
    """inception_score_star_coco.py:104-105 forms the IS* logits from the last layer's WEIGHT MATRIX only
    (``w = ...("softmax/logits/MatMul").inputs[1]; logits = tf.matmul(tf.squeeze(pool3), w)``).  The stand-in classifier
    bias is non-zero by construction, so the two heads give clearly different scores: the device path must land on the
    bias-free oracle for --rule coco, on the biased one with --fc-bias on / the bird and ois rules, and the test fails
    for a product that adds the bias under the coco rule (round 4's behaviour)."""
    from PIL import Image
    from tise_toolbox_amd import inception_score as isc, img_data
    from tise_toolbox_amd.inception import fc_bias_for_rule
    assert float(setup["sd"]["fc.bias"].abs().max()) > 0.1
    assert fc_bias_for_rule("coco") is False and fc_bias_for_rule("bird") is True and fc_bias_for_rule("ois") is True
    assert fc_bias_for_rule("coco", "on") is True and fc_bias_for_rule("ois", "off") is False
    d = tmp_path / "imgs"
    d.mkdir()
    for i in range(30):
        Image.fromarray(setup["gen"][i]).save(d / f"{i:05d}.png")
    files = img_data.get_filenames(str(d))
    idx = [int(os.path.basename(f).split(".")[0]) for f in files]
    want_nobias = is_oracle.inception_score_from_logits(setup["lg"][idx], is_oracle.T_COCO, 10, "coco", dtype=np.float32)
    want_bias = is_oracle.inception_score_from_logits(setup["lg_bias"][idx], is_oracle.T_COCO, 10, "coco", dtype=np.float32)
    assert abs(want_nobias[0] - want_bias[0]) > 1e-3, "the stand-in bias does not separate the two heads: the test would test nothing"
    base = ["--image_folder", str(d), "--batch-size", "7", "--synthetic-weights"]
    auto = isc.main(base)
    on = isc.main(base + ["--fc-bias", "on"])
    off = isc.main(base + ["--fc-bias", "off"])
    print("IS* coco head: no bias", auto, "oracle", want_nobias, "| with bias", on, "oracle", want_bias)
    assert abs(auto[0] - want_nobias[0]) <= 1e-4 and abs(auto[1] - want_nobias[1]) <= 1e-4
    assert off == auto
    assert abs(on[0] - want_bias[0]) <= 1e-4 and abs(on[1] - want_bias[1]) <= 1e-4
    assert abs(auto[0] - want_bias[0]) > 1e-3
    # the engine follows the rule of begin(): logits of the same images with and without the bias
    eng, dev = setup["eng"], setup["dev"]
    x = torch.as_tensor(setup["gen"][:8], device=dev)
    eng.begin(n_total=8, rule="bird", temperature=is_oracle.T_BIRD)
    lb = eng.features_from_u8(x)[1].cpu().numpy()
    eng.begin(n_total=8, rule="coco")
    ln = eng.features_from_u8(x)[1].cpu().numpy()
    assert np.abs(lb - setup["lg_bias"][:8]).max() <= 2e-3 * max(1.0, np.abs(setup["lg_bias"]).max())
    assert np.abs(ln - setup["lg"][:8]).max() <= 2e-3 * max(1.0, np.abs(setup["lg"]).max())
    np.testing.assert_allclose(lb - ln, np.broadcast_to(setup["sd"]["fc.bias"].numpy(), lb.shape), atol=1e-4)


def test_object_centric_inception_score(setup, tmp_path):
    """O-IS drop-in (object_fidelity/O-IS/object_centric_inception_score.py): 80-class head, [-1,1] input,
    T = 2.1737587451934814, N // splits rows per split -- against the oracle on the same crops and weights."""
    from PIL import Image
    from tise_toolbox_amd import object_centric_inception_score as ois
    from tise_toolbox_amd.inception import build_inception3
    d = tmp_path / "crops"
    d.mkdir()
    rng = np.random.default_rng(0)
    sizes = [(64, 48), (120, 90), (33, 71)]
    for i in range(37):                                                   # ragged crop sizes, 37 = 3*10 + 7 (tail dropped)
        h, w = sizes[i % 3]
        Image.fromarray(setup["gen"][i % N_GEN][:h, :w]).save(d / f"img{i:03d}_person_{i}.png")
    ds = ois.IgnoreLabelDataset(str(d))
    assert len(ds) == 37 and ds[0].dtype == torch.uint8
    out = tmp_path / "ois.txt"
    mean, std = ois.main(["--image_dir", str(d), "--saved_file", str(out), "--gpu_id", "0", "--synthetic-weights"])
    assert out.read_text() == f"O-IS: {mean} +-  {std}" + SYNTHETIC_TAG
    with pytest.raises(RuntimeError, match="no parameters for 80-class"):
        ois.main(["--image_dir", str(d), "--saved_file", str(out), "--gpu_id", "0"])
    # oracle: PIL-exact resize, (x - 0.5) / 0.5, CPU fp32 trunk without the inception.py:120-124 affine, 80-class fc
    sd = {k: v.float() for k, v in build_inception3(num_classes=80, seed=0, calibration="pm1").state_dict().items()}
    logits = []
    for name in ds.namelist:
        im = np.asarray(Image.open(d / name).convert("RGB"))
        x = resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(im, 299, 299))
        x = (x - np.float32(0.5)) / np.float32(0.5)
        o = inception_oracle.inception_forward(sd, torch.from_numpy(x[None]), resize_input=False, normalize_input=False)[3]
        logits.append(inception_oracle.logits_from_pool3(sd, o).numpy())
    want = is_oracle.inception_score_from_logits(np.concatenate(logits), is_oracle.T_OIS, 10, "ois", dtype=np.float64)
    assert abs(mean - want[0]) <= 1e-4 and abs(std - want[1]) <= 1e-4
    with pytest.raises(AssertionError):
        ois.inception_score(ds, batch_size=64)                            # N > batch_size (reference :26)


def test_hipgraph_replay_matches_eager(cuda_device, monkeypatch):
    """TISE_GRAPH=1: resize + trunk captured once per batch shape and replayed; features and logits must be bit-identical
    to eager launching, for every batch after the capture and for a second batch shape."""
    from tise_toolbox_amd.engine import RealismEngine
    imgs = torch.from_numpy(_cases.smooth_images(40, 256, 256, seed=21)).to(cuda_device)
    monkeypatch.setenv("TISE_GRAPH", "0")
    eager = RealismEngine(dims=2048, seed=0, with_logits=True)
    monkeypatch.setenv("TISE_GRAPH", "1")
    graphed = RealismEngine(dims=2048, seed=0, with_logits=True)
    assert graphed._graph_ok and not eager._graph_ok
    for rep in range(5):                                   # batches 0,1 eager (cache warm-up), 2 captured, 3,4 replayed
        for lo, hi in ((0, 16), (16, 24)):                 # two batch shapes -> two graphs
            batch = imgs[lo + rep:hi + rep]
            f0, l0 = eager.features_from_u8(batch)
            f1, l1 = graphed.features_from_u8(batch)
            assert torch.equal(f0, f1) and torch.equal(l0, l1), (rep, lo)
    assert sum(1 for k in graphed._graphs if not (isinstance(k[0], str))) == 2


def test_u8_stem_path_bit_identical_to_fp32_input_path(cuda_device, monkeypatch):
    """TISE_U8_STEM (default on for the all-HIP trunk): resize writes uint8 only and the stem conv applies the input
    table.  With the fp32-FMA stem kernel (TISE_STEM=fma) features and logits equal the fp32-input path bit for bit (the
    same arithmetic on the same values); the default stem kernel runs the layer on the matrix cores as a split-precision
    K-step (round 3), so it agrees like any other layer does: to ~1e-6 of the scale per layer, here <= 2e-5 of the feature
    and logit scales at the end of the trunk."""
    from tise_toolbox_amd.engine import RealismEngine
    imgs = torch.from_numpy(_cases.smooth_images(24, 256, 256, seed=33)).to(cuda_device)
    monkeypatch.setenv("TISE_U8_STEM", "0")
    a = RealismEngine(dims=2048, seed=0, with_logits=True)
    monkeypatch.setenv("TISE_U8_STEM", "1")
    monkeypatch.setenv("TISE_STEM", "fma")
    b = RealismEngine(dims=2048, seed=0, with_logits=True)
    monkeypatch.delenv("TISE_STEM")
    c = RealismEngine(dims=2048, seed=0, with_logits=True)
    assert b._u8_stem and c._u8_stem and not a._u8_stem and c.fused.stem_mfma and not b.fused.stem_mfma
    fa, la = a.features_from_u8(imgs)
    fb, lb = b.features_from_u8(imgs)
    fc, lc = c.features_from_u8(imgs)
    assert torch.equal(fa, fb) and torch.equal(la, lb)
    assert (fc - fa).abs().max().item() <= 2e-5 * fa.abs().max().item()
    assert (lc - la).abs().max().item() <= 2e-5 * la.abs().max().item()


def _run_ranks(world, argv, tmp_path, module="tise_toolbox_amd.fid_score", timeout=600):
    """`world` processes of a CLI on THIS box's single GPU, rendezvous over gloo (TISE_DIST_BACKEND): the
    data-parallel product code path (shard_files / shard_range + all_reduce of the device buffers) end to end.
    On an 8-GPU node the same code runs with backend nccl (= RCCL)."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    procs = []
    from tise_toolbox_amd.hostinfo import usable_cpus
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), TISE_DIST_BACKEND="gloo", PYTHONPATH=root,
                   OMP_NUM_THREADS=str(max(1, usable_cpus() // world)))
        procs.append(subprocess.Popen([sys.executable, "-m", module] + argv, env=env, cwd=root,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, out))
    return outs


def test_two_ranks_with_an_empty_shard(setup, tmp_path):
    """fewer batches than ranks.  5 images, batch 2 -> 2 batches; with 3 ranks rank 2 gets nothing and must
    still join the all-reduce; the result equals the single-process run.  And N < batch globally: every rank raises
    the same ZeroDivisionError instead of one crashing while the others hang in the collective."""
    from PIL import Image
    from tise_toolbox_amd import fid_score
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir(); rdir.mkdir()
    for i in range(5):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:05d}.png")
    for i in range(7):
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:05d}.png")
    argv = ["--batch-size", "2", "--path1", str(rdir), "--path2", str(gdir), "--num-workers", "0", "--synthetic-weights"]
    single = fid_score.main(argv)
    out = tmp_path / "dp.txt"
    res = _run_ranks(3, argv + ["--saved_file", str(out)], tmp_path)
    assert all(rc == 0 for rc, _ in res), res
    got = float(out.read_text().split()[1])
    # N << d: the Frechet distance of two rank-5 covariances; the shards change the fp64 summation order of S
    assert abs(got - single) <= 1e-5, (got, single)
    res = _run_ranks(2, ["--batch-size", "16", "--path1", str(rdir), "--path2", str(gdir), "--num-workers", "0",
                         "--synthetic-weights"], tmp_path, timeout=300)
    assert all(rc != 0 and "ZeroDivisionError" in o for rc, o in res), res


def test_ragged_crops_one_trunk_pass_and_per_class_fid(setup, tmp_path):
    """ADVICE r1 + section 8 f2: crops of different sizes are resized into ONE batch (a single trunk pass), results
    equal the per-image path bit for bit; --per-class groups crops by the {class} token (crop_object.py:45) and
    every class's FID equals the plain FID of a directory holding only that class."""
    from PIL import Image
    from tise_toolbox_amd import fid_score
    eng = setup["eng"]
    rng = np.random.default_rng(5)
    crops = [torch.from_numpy(setup["gen"][i][:int(rng.integers(20, 200)), :int(rng.integers(20, 200))].copy()) for i in range(12)]
    fa, la = eng.features_from_u8_list(crops)
    for i in (0, 5, 11):
        fb, lb = eng.features_from_u8(crops[i].unsqueeze(0).to(eng.device))
        # same arithmetic per image; the batch size only changes which workgroup computes a pixel
        assert torch.allclose(fa[i], fb[0], rtol=0, atol=1e-6 * float(fb.abs().max()))
    assert fid_score.class_of_crop("/x/COCO_val_000012_traffic light_37.png") == "traffic light"
    assert fid_score.class_of_crop("img_7_dog_0.png") == "dog"
    with pytest.raises(ValueError):
        fid_score.class_of_crop("plain.png")
    classes = ["dog", "traffic light", "cup"]
    counts = {"gen": {"dog": 6, "traffic light": 5, "cup": 1}, "ref": {"dog": 4, "traffic light": 7, "cup": 3}}
    k = 0
    for side in ("gen", "ref"):
        for c in classes:
            os.makedirs(tmp_path / f"{side}_{c}", exist_ok=True)
        os.makedirs(tmp_path / side, exist_ok=True)
        for j in range(max(counts[side].values())):                      # interleave the classes in the directory
            for c in classes:
                if j >= counts[side][c]:
                    continue
                im = setup[side][k % len(setup[side])][:40 + 9 * (k % 7), :50 + 11 * (k % 5)]
                name = f"im_{k}_{c}_{k}.png"
                Image.fromarray(im).save(tmp_path / side / name)
                Image.fromarray(im).save(tmp_path / f"{side}_{c}" / name)
                k += 1
    out = tmp_path / "pc.txt"
    per = fid_score.main(["--batch-size", "4", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"),
                          "--label", "O-FID", "--num-classes", "80", "--per-class", "--num-workers", "0",
                          "--synthetic-weights", "--saved_file", str(out)])
    assert list(per) == ["dog", "traffic light"]                         # cup: a single crop on the gen side
    text = out.read_text()
    assert "O-FID[dog]: " in text and "skipped" in text and "cup" in text.splitlines()[-1]
    for c in per:
        n1, n2 = counts["ref"][c], counts["gen"][c]
        # plain FID of the class's own directories; batch size = class size so that nothing is dropped
        from tise_toolbox_amd.inception import InceptionV3
        model = InceptionV3([3], num_classes=80, seed=0).cuda()
        m1, s1 = fid_score._compute_statistics_of_path(str(tmp_path / f"ref_{c}"), model, n1, 2048, True, 0)
        m2, s2 = fid_score._compute_statistics_of_path(str(tmp_path / f"gen_{c}"), model, n2, 2048, True, 0)
        want = fid_score.calculate_frechet_distance(m1, s1, m2, s2)
        assert abs(per[c] - want) <= 1e-6 * max(1.0, abs(want)), (c, per[c], want)
    # the per-class solves are SHARDED over the ranks (class i of the sorted union -> rank i mod W: reduce to the owner,
    # solve there, all-reduce of the scalars): 2 and 3 ranks on this GPU reproduce the one-process result per class
    for world in (3,):                                                    # (2 ranks: the same code path, run during development -- a rank costs ~50 s of process start-up)
        outw = tmp_path / f"pc_{world}.txt"
        res = _run_ranks(world, ["--batch-size", "4", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"),
                                 "--label", "O-FID", "--num-classes", "80", "--per-class", "--num-workers", "0",
                                 "--synthetic-weights", "--saved_file", str(outw)], tmp_path)
        assert all(rc == 0 for rc, _ in res), res
        lines = outw.read_text().splitlines()
        got = {ln[len("O-FID["):ln.index("]")]: float(ln.split("]: ")[1].split()[0]) for ln in lines if ln.startswith("O-FID[")}
        assert list(got) == list(per)
        for c in per:
            # N << d: the shards change the fp64 summation order of S (as test_two_ranks_with_an_empty_shard)
            assert abs(got[c] - per[c]) <= 1e-9 * max(1.0, abs(per[c])) + 1e-5, (world, c, got[c], per[c])
        assert "cup" in lines[-1]
    # ... and the CPU ORACLE on the same crops (object_fidelity/O-FID/fid_score.py:188-205 applied to
    # one class's files): PIL-exact resize of every crop, CPU fp32 InceptionV3 with the same 80-class stand-in
    # weights, np.cov, scipy sqrtm.  |dFID| <= 1e-3.
    from tise_toolbox_amd import img_data
    from tise_toolbox_amd.inception import build_inception3
    sd80 = {k: v.float() for k, v in build_inception3(num_classes=80, seed=0).state_dict().items()}

    def oracle_stats(root):
        files = img_data.get_filenames(str(root))
        x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(np.asarray(Image.open(f).convert("RGB")), 299, 299))
                      for f in files])
        act = inception_oracle.inception_forward(sd80, torch.from_numpy(x))[3].flatten(1).numpy().astype(np.float64)
        return fid_oracle.calculate_activation_statistics(act)
    for c in per:
        want = fid_oracle.calculate_frechet_distance(*oracle_stats(tmp_path / f"ref_{c}"), *oracle_stats(tmp_path / f"gen_{c}"))
        print("per-class O-FID", c, "device", per[c], "oracle", want)
        assert abs(per[c] - want) <= 1e-3, (c, per[c], want)


def test_baseline_config0_1k_vs_1k_random_pngs(cuda_device, tmp_path, monkeypatch):
    """BASELINE.json configs[0] / SURVEY 8(d) Config 1 end to end: 1 000 generated + 1 000 reference 256x256 PNGs of
    i.i.d. uniform bytes (default_rng(0) / default_rng(1)), batch 50, through the drop-in CLI -- against the CPU
    oracle on the SAME files in the same walk order (PIL-exact resize, CPU fp32 InceptionV3, np.cov, scipy sqrtm).
    N < d: both covariances have rank <= 999 (the reference's own rank-deficient regime).
    Tolerance: north_star's ABSOLUTE |dFID| <= 1e-3.  (Round 2 asserted a relative bound here because its stand-in
    weights, calibrated on smooth fields only, blew white noise up to FID ~1.3e3; round 3's calibration batch spans
    smooth -> white noise, inception.CALIBRATION_NOISE_FRACTIONS, and this job lands in the published FID range.)
    The same files also go through the exact-fp32 MIOpen trunk (TISE_CONV=miopen): its distance from the oracle is
    printed next to the split-fp16 trunk's, so the share of the operand format in the error is visible
    (tools/config0_floor.py adds the oracle-vs-oracle floor at 1 vs 16 host threads)."""
    from PIL import Image
    from tise_toolbox_amd import fid_score, img_data
    from tise_toolbox_amd.inception import build_inception3
    n = 1000
    for name, seed in (("gen", 0), ("ref", 1)):
        rng = np.random.default_rng(seed)
        os.makedirs(tmp_path / name)
        for i in range(n):
            Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)).save(tmp_path / name / f"{i:05d}.png", compress_level=1)
    out = tmp_path / "fid.txt"
    got = fid_score.main(["--batch-size", "50", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"),
                          "--saved_file", str(out), "--num-workers", "8", "--synthetic-weights"])
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))    # the CPU forward is 3x faster at 16 threads than at 128

    def oracle_stats(root):
        files = img_data.get_filenames(str(root))
        files = files[:fid_oracle.n_used_images(len(files), 50)]
        feats = []
        for i in range(0, len(files), 50):
            x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(
                np.asarray(Image.open(f).convert("RGB")), 299, 299)) for f in files[i:i + 50]])
            feats.append(inception_oracle.inception_forward(sd, torch.from_numpy(x))[3].flatten(1).numpy())
        act = np.concatenate(feats).astype(np.float64)                    # fid_score.py:98 float64 pred_arr
        assert act.shape == (n, 2048)
        return fid_oracle.calculate_activation_statistics(act)
    m1, s1 = oracle_stats(tmp_path / "ref")
    m2, s2 = oracle_stats(tmp_path / "gen")
    want = fid_oracle.calculate_frechet_distance(m1, s1, m2, s2)
    monkeypatch.setenv("TISE_CONV", "miopen")
    monkeypatch.setenv("TISE_MIOPEN_FIND", "0")
    got32 = fid_score.main(["--batch-size", "50", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"),
                            "--num-workers", "8", "--synthetic-weights"])
    print("config0 FID split-fp16 trunk", got, "MIOpen-fp32 trunk", got32, "oracle", want,
          "|split - oracle|", abs(got - want), "|miopen - oracle|", abs(got32 - want))
    assert 2.0 <= want <= 200.0, want                      # the published range the absolute budget is meant for
    assert abs(got - want) <= 1e-3, (got, want)
    assert abs(got32 - want) <= 1e-3, (got32, want)


def test_config0_style_point_near_the_top_of_the_published_range(cuda_device, tmp_path):
    """test_baseline_config0... samples the published FID range (2 .. 200, README.md:487-497) at its bottom
    (3.5).  Second point near the TOP: 300 generated white-noise PNGs (default_rng(0)) against 300 reference PNGs of
    white noise at 0.47 of the contrast (default_rng(1): byte * 0.47 + 68), batch 50, through the drop-in CLI, against the
    CPU oracle on the same files.  The oracle lands at FID ~180; the same ABSOLUTE bound |dFID| <= 1e-3."""
    from PIL import Image
    from tise_toolbox_amd import fid_score, img_data
    from tise_toolbox_amd.inception import build_inception3
    n = 300
    for name, seed in (("gen", 0), ("ref", 1)):
        rng = np.random.default_rng(seed)
        os.makedirs(tmp_path / name)
        for i in range(n):
            im = rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)
            if name == "ref":
                im = (im.astype(np.float32) * 0.47 + 68).astype(np.uint8)
            Image.fromarray(im).save(tmp_path / name / f"{i:05d}.png", compress_level=1)
    got = fid_score.main(["--batch-size", "50", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"),
                          "--num-workers", "8", "--synthetic-weights"])
    sd = {k: v.float() for k, v in build_inception3(seed=0).state_dict().items()}
    torch.set_num_threads(min(16, torch.get_num_threads()))

    def oracle_stats(root):
        files = img_data.get_filenames(str(root))
        files = files[:fid_oracle.n_used_images(len(files), 50)]
        feats = []
        for i in range(0, len(files), 50):
            x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(
                np.asarray(Image.open(f).convert("RGB")), 299, 299)) for f in files[i:i + 50]])
            feats.append(inception_oracle.inception_forward(sd, torch.from_numpy(x))[3].flatten(1).numpy())
        return fid_oracle.calculate_activation_statistics(np.concatenate(feats).astype(np.float64))
    want = fid_oracle.calculate_frechet_distance(*oracle_stats(tmp_path / "ref"), *oracle_stats(tmp_path / "gen"))
    print("config0-style top-of-range point: device", got, "oracle", want, "|d|", abs(got - want))
    assert 150.0 <= want <= 200.0, want
    assert abs(got - want) <= 1e-3, (got, want)


def test_device_batch_is_decoupled_from_batch_size(setup, tmp_path, monkeypatch):
    """--batch-size defines the drop-last rule and the shard borders (fid_score.py:90-96,215-217), NOT
    the size of a trunk pass: the loader's batches are gathered into device batches of up to TISE_DEVICE_BATCH images
    (engine.device_batch_images / coalesce_u8; img_data.U8CacheLoader(group=K)).  Features are bit-identical whatever the
    grouping; the FID moves only by the fp64 summation order of S."""
    from PIL import Image
    from tise_toolbox_amd import engine, fid_score
    from tise_toolbox_amd.inception import InceptionV3
    assert engine.device_batch_images(50) == 1000 and engine.device_batch_images(64) == 960
    assert engine.device_batch_images(3000) == 3000 and engine.device_batch_images(50, 4096 * 4096 * 3) == 50
    dev = setup["dev"]
    model = InceptionV3([3], seed=0).cuda()
    batches = [torch.from_numpy(setup["gen"][i:i + 4]) for i in range(0, 44, 4)]          # 11 host batches of 4
    passes = []
    orig = engine.RealismEngine.features_from_u8

    def spy(self, b):
        passes.append(int(b.shape[0]))
        return orig(self, b)
    monkeypatch.setattr(engine.RealismEngine, "features_from_u8", spy)
    acts = {}
    for limit in ("4", "12", "1000"):
        monkeypatch.setenv("TISE_DEVICE_BATCH", limit)
        passes.clear()
        acts[limit] = fid_score.get_activations(batches, model, batch_size=4, dims=2048, cuda=True, verbose=False)
        assert passes == {"4": [4] * 11, "12": [12, 12, 12, 8], "1000": [44]}[limit], (limit, passes)
    assert acts["4"].shape == (44, 2048) and np.array_equal(acts["4"], acts["12"]) and np.array_equal(acts["4"], acts["1000"])
    # a device batch in the middle of the list keeps its place; ragged lists pass through in order
    mixed = [batches[0], batches[1].to(dev), [torch.from_numpy(setup["gen"][8][:100, :90].copy()), torch.from_numpy(setup["gen"][9])], batches[3]]
    got = list(engine.coalesce_batches(iter(mixed), dev, 1000))
    assert [len(g) for g in got] == [8, 2, 4] and isinstance(got[1], list)
    assert torch.equal(got[0].cpu(), torch.cat([batches[0], batches[1]])) and torch.equal(got[2].cpu(), batches[3])
    # the CLI, PNG and --u8-cache feeds: 23 generated / 17 reference images at batch 4 -> 20 / 16 used
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir(); rdir.mkdir()
    for i in range(23):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:05d}.png")
    for i in range(17):
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:05d}.png")
    argv = ["--batch-size", "4", "--path1", str(rdir), "--path2", str(gdir), "--num-workers", "2", "--synthetic-weights"]
    fids = {}
    for limit in ("4", "12", "1000"):
        monkeypatch.setenv("TISE_DEVICE_BATCH", limit)
        passes.clear()
        fids[limit, "png"] = fid_score.main(argv)
        assert sum(passes) == 36 and max(passes) == min(int(limit), 20), (limit, passes)
        passes.clear()
        fids[limit, "u8"] = fid_score.main(argv + ["--u8-cache"])
        assert sum(passes) == 36 and max(passes) == min(int(limit), 20), (limit, passes)
    ref_fid = fids["4", "png"]
    print("FID by device batch / feed:", fids)
    for k, v in fids.items():
        assert abs(v - ref_fid) <= 1e-6, (k, v, ref_fid)        # N << d: only the summation order of S differs


def test_split_trunk_vs_exact_fp32_convs_3000_images(cuda_device, monkeypatch):
    """Mid-size cross-check that needs no CPU: 3 000 bench-style images through the split-fp16 trunk and through
    MIOpen's exact-fp32 convolutions (same weights, same statistics / Frechet / IS* kernels).  N > d: both
    covariances are full rank.  |dFID| <= 1e-3, |dIS| <= 1e-4 (north_star), feature error <= 1e-4 of the scale."""
    import bench
    from tise_toolbox_amd.engine import RealismEngine, T_COCO, frechet_solver
    monkeypatch.setenv("TISE_CONV", "split")
    eng = RealismEngine(dims=2048, seed=0, with_logits=True)
    n, B = 3000, 250
    data = torch.cat([bench.synth_images_device(i, i + 500, cuda_device, seed=0) for i in range(0, n, 500)])
    eng.begin(n_total=n)
    for i in range(0, n, B):
        eng.step_u8(bench.synth_images_device(i, i + B, cuda_device, seed=1, shift=0.12), i)
    mu_ref, sigma_ref = eng.statistics()
    eng.begin(n_total=n, temperature=T_COCO)
    chunks = [(a, a + B) for a in range(0, n, B)]
    for a, b in chunks:
        eng.step_u8(data[a:b], a)
    mu, sigma = eng.statistics()
    solver = frechet_solver(2048, cuda_device)
    fid = float(solver.distance(mu, sigma, mu_ref, sigma_ref)["fid"])
    cc = bench.cross_check_fp32(eng, data, chunks, 0, n, mu, sigma, mu_ref, sigma_ref, fid, eng.inception_score(),
                                solver, cuda_device)
    print(cc)
    assert cc["dfid"] <= 1e-3 and cc["dis"] <= 1e-4 and cc["dis_std"] <= 1e-4
    assert cc["max_feature_err_rel"] <= 1e-4


def test_u8_cache_feed_is_bit_identical_to_png_decoding(setup, tmp_path):
    """--u8-cache: first run decodes into <dir>/.tise_u8_cache.npy, later runs read it through the
    double-buffered pinned host->device loader; same FID to the last bit as the DataLoader path, the cache file is
    never mistaken for an image, and a changed directory invalidates it."""
    from PIL import Image
    from tise_toolbox_amd import fid_score, img_data
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir(); rdir.mkdir()
    for i in range(23):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:05d}.png")
    for i in range(17):
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:05d}.png")
    argv = ["--batch-size", "4", "--path1", str(rdir), "--path2", str(gdir), "--num-workers", "2", "--synthetic-weights"]
    plain = fid_score.main(argv)
    first = fid_score.main(argv + ["--u8-cache"])                     # builds both caches
    assert (gdir / fid_score.U8_CACHE_NAME).exists() and (rdir / fid_score.U8_CACHE_NAME).exists()
    assert np.load(gdir / fid_score.U8_CACHE_NAME, mmap_mode="r").shape == (23, 256, 256, 3)
    again = fid_score.main(argv + ["--u8-cache"])                     # reads them
    assert plain == first == again
    assert len(img_data.get_filenames(str(gdir))) == 23               # the cache is not walked as an image
    Image.fromarray(setup["gen"][30]).save(gdir / "extra.png")        # directory changed -> cache rebuilt
    changed = fid_score.main(argv + ["--u8-cache"])
    assert np.load(gdir / fid_score.U8_CACHE_NAME, mmap_mode="r").shape[0] == 24
    assert changed == fid_score.main(argv)
    loader = img_data.U8CacheLoader(str(gdir / fid_score.U8_CACHE_NAME), 5, setup["dev"], rows=(5, 20))
    got = torch.cat([b.clone() for b in loader])
    assert len(loader) == 3 and torch.equal(got.cpu(), torch.from_numpy(np.load(gdir / fid_score.U8_CACHE_NAME)[5:20]))


def test_png_ring_feed_is_bit_identical_to_the_dataloader_feed(setup, tmp_path, capfd):
    """Row a2: --png-feed ring (decode processes -> shared page-locked ring -> side-stream H2D, png_ring.py) against
    --png-feed dataloader (DataLoader workers + collate + pin_memory, the round 1-4 path): same files, same walk order, same
    drop-last rule -> the SAME FID to the last bit, for several worker counts and a device batch that straddles chunks; a
    directory with one image of another size falls back to the DataLoader path and still gives the DataLoader result."""
    from PIL import Image
    from tise_toolbox_amd import fid_score
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir(); rdir.mkdir()
    for i in range(47):
        Image.fromarray(setup["gen"][i % N_GEN]).save(gdir / f"{i:05d}.png")
    for i in range(31):
        Image.fromarray(setup["ref"][i % N_REF]).save(rdir / f"{i:05d}.png")
    base = ["--batch-size", "5", "--path1", str(rdir), "--path2", str(gdir), "--synthetic-weights"]
    want = fid_score.main(base + ["--png-feed", "dataloader", "--num-workers", "2"])
    capfd.readouterr()
    for extra in (["--num-workers", "3"], ["--num-workers", "16"], []):
        got = fid_score.main(base + ["--png-feed", "ring"] + extra)
        err = capfd.readouterr().err
        assert "shared pinned ring" in err and "falling back" not in err, err
        assert got == want, (extra, got, want)
    os.environ["TISE_DEVICE_BATCH"] = "15"                    # device batches of 15 images: chunks of 8 straddle them
    try:                                                      # (another batching of the fp64 sums: compare like with like)
        want15 = fid_score.main(base + ["--png-feed", "dataloader", "--num-workers", "2"])
        assert fid_score.main(base + ["--png-feed", "ring", "--num-workers", "4"]) == want15
        assert abs(want15 - want) <= 1e-5
    finally:
        del os.environ["TISE_DEVICE_BATCH"]
    # statistics-only mode goes through the ring too
    so = tmp_path / "s.npz"
    fid_score.main(["--batch-size", "5", "--path2", str(gdir), "--save-stats", str(so), "--synthetic-weights"])
    assert abs(fid_score.main(["--batch-size", "5", "--path1", str(so), "--path2", str(gdir), "--synthetic-weights"])) <= 1e-4
    # ragged directory: fallback (the odd file must be one of the 45 USED files: the walk order is the file system's, and the
    # drop-last rule cuts its tail -- overwriting a fixed name was in the dropped tail on some boxes)
    from tise_toolbox_amd import img_data
    Image.fromarray(setup["gen"][3][:100, :120]).save(img_data.get_filenames(str(gdir))[7])
    capfd.readouterr()
    rag_dl = fid_score.main(base + ["--png-feed", "dataloader", "--num-workers", "2"])
    rag_ring = fid_score.main(base + ["--png-feed", "ring", "--num-workers", "4"])
    assert "falling back to the DataLoader path" in capfd.readouterr().err
    assert rag_ring == rag_dl


COCO80 = ("person,bicycle,car,motorcycle,airplane,bus,train,truck,boat,traffic light,fire hydrant,stop sign,parking meter,bench,bird,cat,"
          "dog,horse,sheep,cow,elephant,bear,zebra,giraffe,backpack,umbrella,handbag,tie,suitcase,frisbee,skis,snowboard,sports ball,"
          "kite,baseball bat,baseball glove,skateboard,surfboard,tennis racket,bottle,wine glass,cup,fork,knife,spoon,bowl,banana,apple,"
          "sandwich,orange,broccoli,carrot,hot dog,pizza,donut,cake,chair,couch,potted plant,bed,dining table,toilet,tv,laptop,mouse,"
          "remote,keyboard,cell phone,microwave,oven,toaster,sink,refrigerator,book,clock,vase,scissors,teddy bear,hair drier,"
          "toothbrush").split(",")


def test_per_class_o_fid_at_80_classes(setup, tmp_path, monkeypatch):
    """BASELINE configs[4] AT ITS SIZE: 80 classes x 40-48 ragged crops per side through
    ``fid_score --per-class``.  (a) every class against the plain single-class path applied to the very feature rows the CLI
    produced (non-grouped covariance kernel, one accumulator, calculate_frechet_distance): <= 1e-9 -- the grouped launch,
    the class sort and the sharded solves change nothing; (b) a sampled class against the plain CLI run on a directory
    holding only that class: <= 1e-6; (c) five sampled classes against the CPU oracle (PIL-exact resize, CPU fp32
    InceptionV3, np.cov, scipy sqrtm): <= 1e-3; (d) two ranks on this GPU reproduce the per-class values; wall times printed."""
    import time
    from PIL import Image
    from tise_toolbox_amd import device, fid_score, img_data
    from tise_toolbox_amd.inception import InceptionV3, build_inception3
    assert len(COCO80) == 80 and "traffic light" in COCO80 and "hair drier" in COCO80
    sampled_cli, sampled_oracle = ["traffic light"], ["person", "dog", "traffic light", "pizza", "toothbrush"]
    k = 0
    n_crops = {}
    for side in ("gen", "ref"):
        os.makedirs(tmp_path / side)
        per_class = {c: 40 + (i * (3 if side == "gen" else 5)) % 9 for i, c in enumerate(COCO80)}
        n_crops[side] = sum(per_class.values())
        for c in sampled_cli + sampled_oracle:
            os.makedirs(tmp_path / f"{side}_{c}", exist_ok=True)
        for j in range(max(per_class.values())):                          # classes interleaved in the directory
            for i, c in enumerate(COCO80):
                if j >= per_class[c]:
                    continue
                src = setup[side][(k + 3 * i) % len(setup[side])]
                y0, x0 = (7 * i + 3 * j) % 60, (11 * i + 5 * j) % 60
                im = src[y0:y0 + 48 + 9 * ((k + i) % 11), x0:x0 + 40 + 13 * ((k + j) % 9)]
                name = f"im_{k}_{c}_{k}.png"
                Image.fromarray(im).save(tmp_path / side / name)
                if c in sampled_cli or c in sampled_oracle:
                    Image.fromarray(im).save(tmp_path / f"{side}_{c}" / name)
                k += 1
    captured = []
    orig = device.stats_update_grouped

    def spy(accs, feats_sorted, offsets):
        captured.append((feats_sorted.clone(), [int(o) for o in offsets]))
        return orig(accs, feats_sorted, offsets)
    monkeypatch.setattr(device, "stats_update_grouped", spy)
    argv = ["--batch-size", "50", "--path1", str(tmp_path / "ref"), "--path2", str(tmp_path / "gen"), "--label", "O-FID",
            "--num-classes", "80", "--per-class", "--synthetic-weights"]
    t0 = time.perf_counter()
    per = fid_score.main(argv + ["--saved_file", str(tmp_path / "pc.txt")])
    t_cli = time.perf_counter() - t0
    monkeypatch.setattr(device, "stats_update_grouped", orig)
    print(f"per-class O-FID, 80 classes, {n_crops['ref']} + {n_crops['gen']} ragged crops: {t_cli:.2f} s in-process (model build included)")
    assert list(per) == sorted(COCO80) and len(captured) == 2             # ONE grouped launch per directory
    # (a) the plain single-class path on the CLI's own feature rows
    names = sorted(COCO80)
    (f_ref, o_ref), (f_gen, o_gen) = captured
    worst = 0.0
    for i, c in enumerate(names):
        a1, a2 = device.StatsAccumulator(2048, f_ref.device), device.StatsAccumulator(2048, f_ref.device)
        a1.update(f_ref[o_ref[i]:o_ref[i + 1]].contiguous())
        a2.update(f_gen[o_gen[i]:o_gen[i + 1]].contiguous())
        want = fid_score.calculate_frechet_distance(*a1.finalize(), *a2.finalize())
        worst = max(worst, abs(per[c] - want) / max(1.0, abs(want)))
        a1.close(); a2.close()
    print("80 classes, grouped launch vs single-class path on the same rows: worst relative difference", worst)
    assert worst <= 1e-9
    assert len(set(round(v, 6) for v in per.values())) > 70                # the classes really differ
    # (b) the plain CLI on a directory holding only that class
    model = InceptionV3([3], num_classes=80, seed=0).cuda()
    for c in sampled_cli:
        n1 = len(img_data.get_filenames(str(tmp_path / f"ref_{c}")))
        n2 = len(img_data.get_filenames(str(tmp_path / f"gen_{c}")))
        m1, s1 = fid_score._compute_statistics_of_path(str(tmp_path / f"ref_{c}"), model, n1, 2048, True, 0)
        m2, s2 = fid_score._compute_statistics_of_path(str(tmp_path / f"gen_{c}"), model, n2, 2048, True, 0)
        want = fid_score.calculate_frechet_distance(m1, s1, m2, s2)
        assert abs(per[c] - want) <= 1e-6 * max(1.0, abs(want)), (c, per[c], want)
    # (d) two ranks on this GPU (classes owned by rank i mod 2, reduce to the owner, solve there, all-reduce of the scalars)
    t0 = time.perf_counter()
    res = _run_ranks(2, argv + ["--saved_file", str(tmp_path / "pc2.txt")], tmp_path)
    print(f"2 ranks on one GPU: {time.perf_counter() - t0:.1f} s wall incl. two process start-ups")
    assert all(rc == 0 for rc, _ in res), res
    got = {ln[len("O-FID["):ln.index("]")]: float(ln.split("]: ")[1].split()[0])
           for ln in (tmp_path / "pc2.txt").read_text().splitlines() if ln.startswith("O-FID[")}
    assert list(got) == list(per)
    for c in per:
        assert abs(got[c] - per[c]) <= 1e-9 * max(1.0, abs(per[c])) + 1e-5, (c, got[c], per[c])
    # (c) the CPU oracle on five classes
    sd80 = {k_: v.float() for k_, v in build_inception3(num_classes=80, seed=0).state_dict().items()}

    def oracle_stats(root):
        files = img_data.get_filenames(str(root))
        x = np.stack([resize_oracle.to_tensor(resize_oracle.resize_bilinear_u8(np.asarray(Image.open(f).convert("RGB")), 299, 299))
                      for f in files])
        act = np.concatenate([inception_oracle.inception_forward(sd80, torch.from_numpy(x[i:i + 16]))[3].flatten(1).numpy()
                              for i in range(0, len(x), 16)]).astype(np.float64)
        return fid_oracle.calculate_activation_statistics(act)
    t0 = time.perf_counter()
    for c in sampled_oracle:
        want = fid_oracle.calculate_frechet_distance(*oracle_stats(tmp_path / f"ref_{c}"), *oracle_stats(tmp_path / f"gen_{c}"))
        print("per-class O-FID", c, "device", per[c], "oracle", want)
        assert abs(per[c] - want) <= 1e-3, (c, per[c], want)
    print(f"CPU oracle on five classes: {time.perf_counter() - t0:.1f} s")


@pytest.mark.timeout(600)
def test_range_guard_hit_finishes_the_cli_on_the_exact_path(setup, tmp_path, monkeypatch, capfd):
    """weights whose activations leave the fp16 range of the split format (the stand-in checkpoint with one
    BatchNorm scale blown up) -- the CLI's split-fp16 run raises the range guard, and ``fid_score.main`` finishes the job in
    the same process on the exact-fp32 convolution path; ``--conv exact`` gives the same number directly."""
    from PIL import Image
    from tise_toolbox_amd import fid_score
    monkeypatch.setenv("TISE_MIOPEN_FIND", "0")                       # immediate-mode MIOpen: no solver search in a test
    monkeypatch.setenv("TISE_CONV", "split")                          # (registers the variable for the teardown: the fallback sets it)
    sd = {k: v.clone() for k, v in setup["sd"].items()}
    sd["Conv2d_2b_3x3.bn.weight"] = sd["Conv2d_2b_3x3.bn.weight"] * 3.0e4    # 147 x 147 x 64 activations far beyond 65504
    ck = tmp_path / "blown.pth"
    torch.save(sd, ck)
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir(); rdir.mkdir()
    for i in range(12):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:03d}.png")
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:03d}.png")
    argv = ["--batch-size", "4", "--path1", str(rdir), "--path2", str(gdir), "--weights", str(ck)]
    got = fid_score.main(argv)
    err = capfd.readouterr().err
    assert "again on the exact-fp32 convolution path" in err, err[-500:]
    assert np.isfinite(got)
    monkeypatch.setenv("TISE_CONV", "split")
    exact = fid_score.main(argv + ["--conv", "exact"])
    assert "again on the exact" not in capfd.readouterr().err
    assert abs(got - exact) <= 1e-6 * max(1.0, abs(exact)), (got, exact)


def test_repeated_fid_calls_in_one_process_release_their_model(setup, tmp_path, monkeypatch):
    """The reference's calculate_fid_given_paths builds a model per call and lets it die on return (fid_score.py:229-238).  Here the
    engine hangs on the model and holds it -- a reference cycle that kept every call's device memory (weights, packed weights, statistics
    and staging buffers: ~1.4 GiB at 256 x 256 inputs) until the garbage collector's next full pass; tools/soak_cli_loop.py saw 16 calls
    hold 20 GiB.  With TISE_RELEASE_MODEL=1 fid_score._own_model cuts the cycle on the way out: device memory after the fourth call is
    what it was after the second, with the collector switched off, and the value repeats to the bit.  (Opt-in: fid_score._own_model.)"""
    import gc
    monkeypatch.setenv("TISE_RELEASE_MODEL", "1")
    from PIL import Image
    from tise_toolbox_amd import fid_score
    gdir, rdir = tmp_path / "gen", tmp_path / "ref"
    gdir.mkdir()
    rdir.mkdir()
    for i in range(20):
        Image.fromarray(setup["gen"][i]).save(gdir / f"{i:05d}.png")
    for i in range(15):
        Image.fromarray(setup["ref"][i]).save(rdir / f"{i:05d}.png")
    argv = ["--batch-size", "5", "--path1", str(rdir), "--path2", str(gdir), "--num-workers", "2", "--synthetic-weights"]
    gc.collect()
    gc.disable()
    try:
        vals, mem = [], []
        for _ in range(4):
            vals.append(fid_score.main(argv))
            torch.cuda.synchronize()
            mem.append(torch.cuda.memory_allocated())
    finally:
        gc.enable()
    assert len(set(vals)) == 1, vals
    assert mem[3] <= mem[1] + (8 << 20), mem                      # nothing of a finished call stays (8 MiB of slack for cached scalars)
