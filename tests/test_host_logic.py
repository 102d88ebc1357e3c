"""CPU: host-side logic that mirrors the reference's conventions (file walking, drop-last, sharding,
lookup table, CLI surface, state_dict compatibility)."""
import os

import numpy as np
import pytest
import torch

from oracle import fid_oracle, resize_oracle


def test_get_filenames_rule(tmp_path):
    """img_data.py:27-35: substring match on 'jpg' / 'png' anywhere in the NAME, recursive, os.walk order."""
    from tise_toolbox_amd import img_data
    for name in ["a.png", "b.jpg", "c.jpeg", "d.png.txt", "e.PNG", "f.txt", "sub/g.png", "jpg_notes.md"]:
        p = tmp_path / name
        p.parent.mkdir(parents=True, exist_ok=True)
        p.write_bytes(b"x")
    (tmp_path / "dir.png").mkdir()                     # directories are not files (:33)
    got = img_data.get_filenames(str(tmp_path))
    want = []
    for path, subdirs, files in os.walk(str(tmp_path)):
        for name in files:
            if name.rfind("jpg") != -1 or name.rfind("png") != -1:
                want.append(os.path.join(path, name))
    assert got == want
    base = sorted(os.path.relpath(p, tmp_path) for p in got)
    assert base == ["a.png", "b.jpg", "d.png.txt", "jpg_notes.md", "sub/g.png"]


def test_drop_last_and_sharding():
    from tise_toolbox_amd import dist as tdist
    for n, bs in [(30000, 50), (1000, 64), (37, 5), (3, 5)]:
        assert tdist.n_used_images(n, bs) == fid_oracle.n_used_images(n, bs)
    for n in (0, 1, 7, 600, 601):
        for world in (1, 2, 3, 8):
            ranges = [tdist.shard_range(n, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(ranges[:-1], ranges[1:]))
            sizes = [hi - lo for lo, hi in ranges]
            assert max(sizes) - min(sizes) <= 1
    assert [tdist.shard_range(600, r, 8) for r in range(8)][3] == (225, 300)      # 30k / bs 50 / 8 GPUs


def test_lut_matches_reference_op_order():
    """ToTensor (/255, fp32) then inception.py:120-124 (fp32 mul, fp32 add) evaluated by torch itself."""
    from tise_toolbox_amd import device
    v = torch.arange(256, dtype=torch.uint8)
    x = v.float().div(255)                                              # torchvision ToTensor
    x = x.view(1, 1, 1, 256).repeat(1, 3, 1, 1).clone()
    x[:, 0] = x[:, 0] * (0.229 / 0.5) + (0.485 - 0.5) / 0.5
    x[:, 1] = x[:, 1] * (0.224 / 0.5) + (0.456 - 0.5) / 0.5
    x[:, 2] = x[:, 2] * (0.225 / 0.5) + (0.406 - 0.5) / 0.5
    lut = device.make_lut(True)
    np.testing.assert_array_equal(lut, x[0, :, 0, :].numpy())
    np.testing.assert_array_equal(device.make_lut(False)[1], v.float().div(255).numpy())
    t = torch.from_numpy(np.arange(256, dtype=np.float32) / np.float32(255))
    np.testing.assert_array_equal(device.make_lut(False, scale_pm1=True)[2], ((t - 0.5) / 0.5).numpy())
    np.testing.assert_array_equal(resize_oracle.normalize_input(np.tile(lut_in(), (1, 1)))[...], lut_norm(lut))


def lut_in():
    return (np.arange(256, dtype=np.float32) / np.float32(255)).reshape(1, 1, 256).repeat(3, 0)


def lut_norm(lut):
    return lut.reshape(3, 1, 256)


def test_state_dict_is_torchvision_compatible():
    """Key names / shapes of torchvision.models.inception_v3 (the file the reference downloads)."""
    from tise_toolbox_amd.inception import Inception3
    sd = Inception3().state_dict()
    assert sum(v.numel() for k, v in sd.items() if "num_batches_tracked" not in k and "running" not in k) == 27161264
    for k, shape in {"Conv2d_1a_3x3.conv.weight": (32, 3, 3, 3), "Conv2d_1a_3x3.bn.running_var": (32,),
                     "Mixed_5b.branch5x5_2.conv.weight": (64, 48, 5, 5), "Mixed_6a.branch3x3.conv.weight": (384, 288, 3, 3),
                     "Mixed_6b.branch7x7_2.conv.weight": (128, 128, 1, 7), "Mixed_6e.branch7x7dbl_5.conv.weight": (192, 192, 1, 7),
                     "AuxLogits.conv1.conv.weight": (768, 128, 5, 5), "AuxLogits.fc.weight": (1000, 768),
                     "Mixed_7a.branch7x7x3_4.conv.weight": (192, 192, 3, 3), "Mixed_7c.branch3x3dbl_3b.conv.weight": (384, 384, 3, 1),
                     "fc.weight": (1000, 2048), "fc.bias": (1000,)}.items():
        assert tuple(sd[k].shape) == shape, k
    sd80 = Inception3(num_classes=80).state_dict()                       # O-FID/inception.py:58-64
    assert tuple(sd80["fc.weight"].shape) == (80, 2048) and tuple(sd80["AuxLogits.fc.weight"].shape) == (80, 768)


def test_wrapper_matches_reference_interface_and_oracle():
    from oracle import inception_oracle
    from tise_toolbox_amd.inception import InceptionV3, build_inception3
    assert InceptionV3.BLOCK_INDEX_BY_DIM == {64: 0, 192: 1, 768: 2, 2048: 3} and InceptionV3.DEFAULT_BLOCK_INDEX == 3
    with pytest.raises(AssertionError):
        InceptionV3([4])
    m = InceptionV3([0, 1, 2, 3], seed=0)
    assert not any(p.requires_grad for p in m.parameters())
    x = torch.rand(2, 3, 64, 80)                                         # resize_input path (align_corners=True upsample)
    outs = m(x)
    assert [tuple(o.shape) for o in outs] == [(2, 64, 73, 73), (2, 192, 35, 35), (2, 768, 17, 17), (2, 2048, 1, 1)]
    sd = build_inception3(seed=0).state_dict()
    ref = inception_oracle.inception_forward(sd, x)
    for a, b in zip(outs, ref):
        assert (a - b).abs().max().item() <= 1e-4 * max(1.0, b.abs().max().item())
    only2 = InceptionV3([2], seed=0)
    assert len(only2.blocks) == 3 and tuple(only2(x)[0].shape) == (2, 768, 17, 17)
    m.fold_bn()
    for a, b in zip(m(x), ref):
        assert (a - b).abs().max().item() <= 1e-3 * max(1.0, b.abs().max().item())


def test_cli_surface():
    from tise_toolbox_amd import fid_score, inception_score
    a = fid_score._build_parser().parse_args(["--path1", "a.npz", "--path2", "imgs", "--batch-size", "50", "-c", "3",
                                              "--dims", "768", "--saved_file", "o.txt"])
    assert (a.batch_size, a.dims, a.gpu, a.path1, a.path2, a.saved_file) == (50, 768, "3", "a.npz", "imgs", "o.txt")
    assert fid_score._build_parser().get_default("batch_size") == 64 and fid_score._build_parser().get_default("dims") == 2048
    with pytest.raises(SystemExit):
        fid_score._build_parser().parse_args(["--dims", "100", "--path1", "a", "--path2", "b"])
    b = inception_score._build_parser().parse_args(["--image_folder", "d", "--saved_file", "s.txt", "--gpu", "1"])
    assert (b.image_folder, b.saved_file, b.gpu, b.splits) == ("d", "s.txt", 1, 10)
    assert b.temperature == 0.9091363549232483


def test_ois_surface(tmp_path):
    """object_centric_inception_score.py: --image_dir/--saved_file/--gpu_id, listdir-ordered dataset, asserts."""
    from PIL import Image
    from tise_toolbox_amd import object_centric_inception_score as ois
    a = ois.parse_args(["--image_dir", "d", "--saved_file", "o.txt", "--gpu_id", "2"])
    assert (a.image_dir, a.saved_file, a.gpu_id) == ("d", "o.txt", 2)
    assert ois.T_OIS == 2.1737587451934814 and ois.DEFAULT_WEIGHTS.endswith("inceptionv3_fine_to_with_80_coco_classes.pth")
    for i in range(3):
        Image.fromarray(np.full((5 + i, 4, 3), i, np.uint8)).save(tmp_path / f"c{i}.png")
    ds = ois.IgnoreLabelDataset(str(tmp_path))
    assert ds.namelist == os.listdir(str(tmp_path)) and len(ds) == 3
    assert ds[0].dtype == torch.uint8 and ds[0].shape[2] == 3
    with pytest.raises(AssertionError):
        ois.inception_score(ds, batch_size=3)          # reference :26 requires N > batch_size


def test_missing_path_raises_like_reference(tmp_path):
    from tise_toolbox_amd import fid_score
    with pytest.raises(RuntimeError, match="Invalid path: "):
        fid_score.calculate_fid_given_paths([str(tmp_path / "nope"), str(tmp_path)], 50, "0", 2048)


def test_dataset_yields_uint8_hwc(tmp_path):
    from PIL import Image
    from tise_toolbox_amd import img_data
    rng = np.random.default_rng(0)
    a = rng.integers(0, 256, (20, 30, 3), dtype=np.uint8)
    Image.fromarray(a).save(tmp_path / "a.png")
    Image.fromarray(a[..., 0]).save(tmp_path / "gray.png")               # gray -> convert("RGB") (img_data.py:21)
    ds = img_data.Dataset(str(tmp_path))
    items = {os.path.basename(f): ds[i] for i, f in enumerate(ds.file_names)}
    np.testing.assert_array_equal(items["a.png"].numpy(), a)
    assert items["gray.png"].shape == (20, 30, 3) and items["gray.png"].dtype == torch.uint8
    np.testing.assert_array_equal(items["gray.png"].numpy()[..., 2], a[..., 0])
    assert isinstance(img_data.collate_u8([items["a.png"], items["gray.png"]]), torch.Tensor)
    assert isinstance(img_data.collate_u8([items["a.png"], items["a.png"][:10]]), list)


# ------------------------------------------------------------------------------------------- RP-COCO host logic (f3)
@pytest.mark.parametrize("name", ["rp_stub_57x10.npz", "rp_stub_40x100.npz"])
def test_rp_host_logic_matches_reference_script_run(golden_dir, name):
    """Bins, success counting, mean / std and the result text of tise_toolbox_amd.RP_coco against what the
    reference script wrote (stub CLIP, tests/golden/make_golden_rp.py); the success flags come from the fixture's
    logits here, from csrc/retrieval.hip in the GPU test."""
    from tise_toolbox_amd import RP_coco
    g = np.load(os.path.join(golden_dir, name))
    success = (np.argmax(g["logits"], axis=1) == 0).astype(np.int64)
    mean, std, scores = RP_coco.r_precision_from_success(success, g["perm"].tolist())
    assert f"R-precision: {mean} +- {std}" == str(g["expected_text"])
    import random
    random.seed(int(g["seed"]))
    ids = list(range(int(g["n_items"])))
    random.shuffle(ids)
    assert ids == g["perm"].tolist() == RP_coco.shuffled_ids(int(g["n_items"]), int(g["seed"]))


def test_rp_caption_table_dedup():
    from tise_toolbox_amd import RP_coco
    items = [{"caption_id": 1, "caption": "a", "mismatched_captions": ["b", "c"]},
             {"caption_id": 2, "caption": "b", "mismatched_captions": ["a", "d"]}]
    caps, idx = RP_coco.caption_table(items)
    assert caps == ["a", "b", "c", "d"] and idx.dtype == np.int32
    assert idx.tolist() == [[0, 1, 2], [1, 0, 3]]
    with pytest.raises(ValueError):
        RP_coco.caption_table(items + [{"caption_id": 3, "caption": "e", "mismatched_captions": ["a"]}])


def test_clip_towers_have_the_published_shape():
    from tise_toolbox_amd import clip_model
    m = clip_model.build_clip(seed=1)
    assert sum(p.numel() for p in m.parameters()) == 151277313          # ViT-B/32 CLIP
    keys = set(m.state_dict())
    for k in ("visual.conv1.weight", "visual.class_embedding", "visual.positional_embedding", "visual.proj",
              "visual.transformer.resblocks.11.attn.in_proj_weight", "visual.transformer.resblocks.0.mlp.c_fc.weight",
              "transformer.resblocks.11.attn.out_proj.bias", "token_embedding.weight", "positional_embedding",
              "ln_final.weight", "text_projection", "logit_scale"):
        assert k in keys, k
    assert m.visual.positional_embedding.shape == (50, 768) and m.positional_embedding.shape == (77, 512)


def test_bpe_tokenizer_on_synthetic_merges(tmp_path):
    import gzip
    from tise_toolbox_amd import clip_model
    merges = ["#version: synthetic", "r e", "re d</w>", "b u", "bu s</w>", "t h", "th e</w>"]
    path = tmp_path / "bpe.txt.gz"
    with gzip.open(path, "wb") as f:
        f.write("\n".join(merges).encode())
    tok = clip_model.BPETokenizer(str(path))
    n_base = 512
    assert tok.encoder["red</w>"] == n_base + 1 and tok.encoder["bus</w>"] == n_base + 3
    ids = tok.encode("The  RED bus")
    assert ids == [tok.encoder["the</w>"], tok.encoder["red</w>"], tok.encoder["bus</w>"]]
    out = tok(["the red bus", "bus"])
    assert out.shape == (2, 77) and out[0, 0] == tok.encoder["<|startoftext|>"] and out[1, 2] == tok.encoder["<|endoftext|>"]
    assert out[0, 5:].sum() == 0
    with pytest.raises(RuntimeError):
        tok(["bus " * 100])


def test_weights_resolution_never_silently_synthetic(tmp_path, monkeypatch, capsys):
    """the reference always loads pretrained parameters; a missing file must not turn into stand-ins."""
    from tise_toolbox_amd import weights
    monkeypatch.setenv("TORCH_HOME", str(tmp_path / "torch_home"))
    monkeypatch.chdir(tmp_path)
    for kind in ("inception", "inception80"):
        with pytest.raises(RuntimeError, match="no parameters for"):
            weights.resolve(None, False, kind)
    path, tag = weights.resolve(None, True, "inception")
    assert path is None and tag == weights.SYNTHETIC_TAG
    assert "SEEDED STAND-IN" in capsys.readouterr().err
    with pytest.raises(RuntimeError, match="Invalid path"):
        weights.resolve(str(tmp_path / "nope.pth"), False, "inception")
    f = tmp_path / "w.pth"; f.write_bytes(b"x")
    assert weights.resolve(str(f), False, "inception") == (str(f), "")
    with pytest.raises(RuntimeError, match="mutually exclusive"):         # an explicit stand-in request is never overridden
        weights.resolve(str(f), True, "inception")
    # the files the reference itself would read are found without a flag (and named on stderr)
    hub = tmp_path / "torch_home" / "hub" / "checkpoints"; hub.mkdir(parents=True)
    (hub / "inception_v3_google-1a9a5a14.pth").write_bytes(b"x")
    capsys.readouterr()
    assert weights.resolve(None, False, "inception") == (str(hub / "inception_v3_google-1a9a5a14.pth"), "")
    assert "inception_v3_google-1a9a5a14.pth" in capsys.readouterr().err
    # ... but --synthetic-weights means the stand-ins on every machine, whatever ~/.cache holds
    assert weights.resolve(None, True, "inception") == (None, weights.SYNTHETIC_TAG)
    (tmp_path / "weights").mkdir()
    (tmp_path / "weights" / "inceptionv3_fine_to_with_80_coco_classes.pth").write_bytes(b"x")
    assert weights.resolve(None, False, "inception80")[0] == os.path.join("weights", "inceptionv3_fine_to_with_80_coco_classes.pth")


def test_ranking_collect_refuses_synthetic_results(tmp_path):
    from tise_toolbox_amd import ranking_score as rs, weights
    f = tmp_path / "fid.txt"
    f.write_text("FID: 12.5")
    assert rs.parse_result_file("FID", str(f)) == 12.5
    f.write_text("FID: 12.5" + weights.SYNTHETIC_TAG)
    with pytest.raises(ValueError, match="synthetic"):
        rs.parse_result_file("FID", str(f))


def test_u8_cache_fingerprint_and_atomic_build(tmp_path):
    """the decoded-pixel cache is valid only for the files it was built from -- images regenerated
    under the same names and count (the evaluate-every-checkpoint workflow) invalidate it -- and a build that dies half
    way leaves nothing that looks like a cache."""
    from PIL import Image
    from tise_toolbox_amd import img_data
    d = tmp_path / "gen"
    d.mkdir()
    rng = np.random.default_rng(0)
    for i in range(6):
        Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).save(d / f"{i}.png")
    files = img_data.get_filenames(str(d))
    cache = str(d / ".tise_u8_cache.npy")
    assert not img_data.u8_cache_is_current(cache, files, str(d))
    img_data.build_u8_cache(files, cache, num_workers=0, root=str(d))
    assert img_data.u8_cache_is_current(cache, files, str(d)) and np.load(cache).shape == (6, 16, 16, 3)
    assert sorted(os.listdir(d)) == sorted([".tise_u8_cache.npy", ".tise_u8_cache.npy.sha256"] + [f"{i}.png" for i in range(6)])
    assert len(img_data.get_filenames(str(d))) == 6                     # neither cache file is walked as an image
    # same names, same count, new pixels
    st = os.stat(d / "3.png")
    Image.fromarray(rng.integers(0, 256, (16, 16, 3), dtype=np.uint8)).save(d / "3.png")
    os.utime(d / "3.png", ns=(st.st_atime_ns, st.st_mtime_ns + 1_000_000))
    assert not img_data.u8_cache_is_current(cache, files, str(d))
    img_data.build_u8_cache(files, cache, num_workers=0, root=str(d))
    assert np.array_equal(np.load(cache)[files.index(str(d / "3.png"))], np.asarray(Image.open(d / "3.png")))
    # a build that fails half way (an image of another size) removes the temporary array AND the old fingerprint
    Image.fromarray(rng.integers(0, 256, (8, 8, 3), dtype=np.uint8)).save(d / "5.png")
    with pytest.raises(ValueError):
        img_data.build_u8_cache(files, cache, num_workers=0, batch_size=2, root=str(d))
    assert not img_data.u8_cache_is_current(cache, files, str(d))
    assert not [n for n in os.listdir(d) if ".tmp" in n] and not os.path.exists(cache + ".sha256")
    # a truncated side file / missing array is "not current", never an exception
    open(cache + ".sha256", "w").write("deadbeef")
    assert not img_data.u8_cache_is_current(cache, files, str(d))


def test_device_batch_rule_and_class_owners(monkeypatch):
    """Round 4 host rules that need no GPU: --batch-size keeps defining the drop-last rule (dist.n_used_images) while a
    trunk pass takes whole loader batches up to TISE_DEVICE_BATCH images and 1 GiB of pixels (engine.device_batch_images);
    per-class O-FID: class i of the sorted list is owned by rank i mod W (dist.class_owners)."""
    from tise_toolbox_amd import dist as tdist
    from tise_toolbox_amd.engine import device_batch_images
    monkeypatch.delenv("TISE_DEVICE_BATCH", raising=False)
    assert device_batch_images(50) == 1000 and device_batch_images(64) == 960 and device_batch_images(1) == 1000
    assert device_batch_images(1000) == 1000 and device_batch_images(1001) == 1001 and device_batch_images(4096) == 4096
    assert device_batch_images(50, 4096 * 4096 * 3) == 50 and device_batch_images(7, 1024 * 1024 * 3) == 336     # 1 GiB cap: 341 images
    monkeypatch.setenv("TISE_DEVICE_BATCH", "120")
    assert device_batch_images(50) == 100 and device_batch_images(64) == 64 and device_batch_images(500) == 500
    # the reference's README recipe: 30 000 images at --batch-size 50 -> nothing dropped, 30 trunk passes instead of 600
    monkeypatch.delenv("TISE_DEVICE_BATCH")
    assert tdist.n_used_images(30000, 50) == 30000 and 30000 // device_batch_images(50) == 30
    assert tdist.n_used_images(30003, 64) == 29952                      # fid_score.py:215-217 drop_last
    names = ["zebra", "cup", "traffic light", "dog", "person"]
    own = tdist.class_owners(names, 3)
    assert own == {"cup": 0, "dog": 1, "person": 2, "traffic light": 0, "zebra": 1}
    assert tdist.class_owners(names, 1) == {c: 0 for c in names} and tdist.class_owners([], 4) == {}
    counts = [sum(1 for c in range(80) if c % 8 == r) for r in range(8)]
    assert counts == [10] * 8                                           # BASELINE configs[4]: 80 classes on 8 GPUs


def test_png_ring_protocol_order_drop_last_and_ragged(tmp_path):
    """png_ring.PngRingLoader without a GPU (iter_host): decode workers write walk-ordered chunks into the shared ring, the
    parent sees every image exactly once and in order whatever the worker / chunk / slot counts; only whole loader batches
    are served (fid_score.py:90-96, 215-217: drop_last); an image of another size is reported as RaggedImages; an unreadable
    file as a RuntimeError naming it; no worker process survives."""
    import subprocess
    from PIL import Image
    from tests import _cases
    from tise_toolbox_amd import img_data, png_ring
    imgs = _cases.smooth_images(45, 64, 48, seed=11)
    files = []
    for i, im in enumerate(imgs):
        p = tmp_path / f"{i:04d}.png"
        (Image.fromarray(im) if i % 7 else Image.fromarray(im).convert("RGBA")).save(p)      # RGBA files: convert("RGB") like img_data.py:21
        files.append(str(p))
    for workers, chunk, bs in ((3, 4, 6), (5, 1, 45), (1, 8, 7), (8, 50, 1)):
        ld = png_ring.PngRingLoader(files, bs, "cpu", workers=workers, chunk=chunk)
        assert len(ld) == 45 // bs
        got = np.zeros((ld.n_rows, 64, 48, 3), np.uint8)
        seen = 0
        procs = list(ld.procs)
        for lo, view in ld.iter_host():
            assert lo == seen
            got[lo:lo + len(view)] = view
            seen += len(view)
        assert seen == (45 // bs) * bs and np.array_equal(got, imgs[:seen])
        assert all(p.poll() is not None for p in procs)
    # the u8 cache is built through the same ring
    cache = str(tmp_path / "cache.npy")
    img_data.build_u8_cache(files, cache, num_workers=3, root=str(tmp_path))
    assert np.array_equal(np.load(cache), imgs)
    Image.fromarray(imgs[0][:30, :20]).save(files[20])
    ld = png_ring.PngRingLoader(files, 5, "cpu", workers=2, chunk=4)
    with pytest.raises(png_ring.RaggedImages, match="0020.png"):
        for _ in ld.iter_host():
            pass
    with pytest.raises(ValueError, match="one size"):
        img_data.build_u8_cache(files, cache, num_workers=2, root=str(tmp_path))
    open(files[20], "wb").write(b"not a png")
    ld = png_ring.PngRingLoader(files, 5, "cpu", workers=2, chunk=4)
    with pytest.raises(RuntimeError, match="png decode worker failed"):
        for _ in ld.iter_host():
            pass
    assert png_ring.auto_workers(1) >= 2 and png_ring.auto_workers(8) >= 2


@pytest.mark.timeout(180)
def test_png_ring_reports_a_killed_worker_instead_of_waiting_for_its_chunk(tmp_path):
    """A decode worker that is killed (out of memory, a signal) while it holds a claimed chunk: the surviving workers fill the
    ring and wait for the consumer, the consumer waits for the chunk nobody will finish.  The loader must raise, not hang."""
    import time
    from PIL import Image
    from tests import _cases
    from tise_toolbox_amd import _png_worker, png_ring
    imgs = _cases.smooth_images(45, 32, 32, seed=12)
    files = []
    for i, im in enumerate(imgs):
        files.append(str(tmp_path / f"{i:04d}.png"))
        Image.fromarray(im).save(files[-1])
    ld = png_ring.PngRingLoader(files, 1, "cpu", workers=2, chunk=2)           # 23 chunks, 8 slots
    ld.start()
    t0 = time.time()
    while int(ld.hdr[_png_worker.HDR_NEXT]) < ld.nslots + 2 and time.time() - t0 < 60:      # nobody consumes: both workers end up
        time.sleep(0.01)                                                               # blocked, each holding a claimed chunk
    assert int(ld.hdr[_png_worker.HDR_NEXT]) == ld.nslots + 2
    ld.procs[0].kill()
    ld.procs[0].wait()
    with pytest.raises(RuntimeError, match="worker died"):
        for _ in ld.iter_host():
            pass
    assert all(p.poll() is not None for p in ld.procs)


def _png_bytes(img, ft, split=1):
    """A PNG file image of `img` (h, w, 3 | 4) whose rows all use filter type `ft` (5: a different one per row), the
    compressed stream cut into `split` IDAT chunks, with an ancillary chunk in front."""
    import struct
    import zlib
    h, w, c = img.shape

    def chunk(t, d):
        return struct.pack(">I", len(d)) + t + d + struct.pack(">I", zlib.crc32(t + d))
    raw = bytearray()
    prev = np.zeros(w * c, np.int32)
    for y in range(h):
        cur = img[y].reshape(-1).astype(np.int32)
        f = ft if ft < 5 else (y * 7 + 3) % 5
        left = np.concatenate([np.zeros(c, np.int32), cur[:-c]])
        ul = np.concatenate([np.zeros(c, np.int32), prev[:-c]])
        if f == 0:
            enc = cur
        elif f == 1:
            enc = cur - left
        elif f == 2:
            enc = cur - prev
        elif f == 3:
            enc = cur - ((left + prev) >> 1)
        else:
            p = left + prev - ul
            pa, pb, pc = abs(p - left), abs(p - prev), abs(p - ul)
            enc = cur - np.where((pa <= pb) & (pa <= pc), left, np.where(pb <= pc, prev, ul))
        raw.append(f)
        raw += (enc & 255).astype(np.uint8).tobytes()
        prev = cur
    z = zlib.compress(bytes(raw), 6)
    parts = [z[i * len(z) // split:(i + 1) * len(z) // split] for i in range(split)]
    return (b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if c == 3 else 6, 0, 0, 0))
            + chunk(b"gAMA", struct.pack(">I", 45455)) + b"".join(chunk(b"IDAT", p) for p in parts) + chunk(b"IEND", b""))


def test_native_png_decoder_equals_pillow(monkeypatch):
    """csrc/png_decode.c against Pillow's Image.open(f).convert("RGB") (img_data.py:19-25): every row filter, RGB and RGBA,
    one and several IDAT chunks, odd sizes, with libdeflate and with zlib; files outside the subset and broken files are
    handed back (non-zero code), never decoded differently."""
    import ctypes
    import io
    import subprocess
    import sys
    from PIL import Image
    from tise_toolbox_amd import _png_worker, build
    build.build_png(force=False, verbose=False)
    lib = _png_worker.load_decoder()
    assert lib is not None

    def dec(lib_, b, h, w):
        out = np.full((h, w, 3), 7, np.uint8)
        n = lib_.tise_png_scratch_bytes(h, w, len(b))
        sc = np.empty(n, np.uint8)
        gw, gh = ctypes.c_int(), ctypes.c_int()
        rc = lib_.tise_png_decode_rgb8(b, len(b), out.ctypes.data, h, w, sc.ctypes.data, n, ctypes.byref(gw), ctypes.byref(gh))
        return rc, out, (gh.value, gw.value)
    rng = np.random.default_rng(0)
    cases = []
    for c in (3, 4):
        for ft in range(6):
            for (h, w) in ((1, 1), (5, 7), (64, 33), (256, 256)):
                img = rng.integers(0, 256, (h, w, c), dtype=np.uint8)
                if ft == 4:
                    img[:, :, 0] = np.clip(np.add.outer(np.arange(h), np.arange(w)) * 2, 0, 255)      # smooth ramps: Paeth ties
                for split in (1, 3):
                    cases.append((_png_bytes(img, ft, split), h, w))
    for b, h, w in cases:
        want = np.asarray(Image.open(io.BytesIO(b)).convert("RGB"))
        rc, out, _ = dec(lib, b, h, w)
        assert rc == _png_worker.PNG_OK and np.array_equal(out, want)
    # Pillow-written files (what the toolbox is fed with), RGB and RGBA
    from tests import _cases
    for im in _cases.smooth_images(6, 96, 80, seed=2):
        for mode in ("RGB", "RGBA"):
            bio = io.BytesIO()
            Image.fromarray(im).convert(mode).save(bio, format="PNG")
            rc, out, _ = dec(lib, bio.getvalue(), 96, 80)
            assert rc == _png_worker.PNG_OK and np.array_equal(out, im)
    base = Image.fromarray(_cases.smooth_images(1, 32, 32, seed=3)[0])
    for variant, fmt, kw in ((base.convert("L"), "PNG", {}), (base.convert("P"), "PNG", {}), (base, "JPEG", {}),
                             (base.convert("I;16"), "PNG", {}), (base, "PNG", {"transparency": (0, 0, 0)})):
        bio = io.BytesIO()
        variant.save(bio, format=fmt, **kw)
        assert dec(lib, bio.getvalue(), 32, 32)[0] == _png_worker.PNG_UNSUPPORTED
    good = cases[-1][0]
    assert dec(lib, good, 100, 100)[0::2] == (_png_worker.PNG_SIZE, (256, 256))
    assert dec(lib, good[:len(good) // 2], 256, 256)[0] == _png_worker.PNG_CORRUPT
    assert dec(lib, b"not a png at all, but long enough to be looked at......", 4, 4)[0] == _png_worker.PNG_UNSUPPORTED
    # the zlib back end (a fresh process: the choice is made once per process)
    code = ("import sys, ctypes, numpy as np; sys.path.insert(0, %r)\n"
            "from tise_toolbox_amd import _png_worker\n"
            "lib = _png_worker.load_decoder(); assert lib.tise_png_inflate_backend() == 0\n"
            "b = open(sys.argv[1], 'rb').read(); out = np.empty((256, 256, 3), np.uint8)\n"
            "n = lib.tise_png_scratch_bytes(256, 256, len(b)); sc = np.empty(n, np.uint8)\n"
            "rc = lib.tise_png_decode_rgb8(b, len(b), out.ctypes.data, 256, 256, sc.ctypes.data, n, None, None)\n"
            "sys.stdout.buffer.write(bytes([rc]) + out.tobytes())\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    import tempfile
    with tempfile.NamedTemporaryFile(suffix=".png") as tf:
        tf.write(good)
        tf.flush()
        r = subprocess.run([sys.executable, "-c", code, tf.name], capture_output=True, env=dict(os.environ, TISE_PNG_ZLIB="1"))
    assert r.returncode == 0 and r.stdout[0] == 0, r.stderr[-500:]
    assert np.array_equal(np.frombuffer(r.stdout[1:], np.uint8).reshape(256, 256, 3), np.asarray(Image.open(io.BytesIO(good)).convert("RGB")))


def test_item_schedule_properties(monkeypatch):
    """engine.item_schedule: the device batches of a fed image set -- whole loader batches, at most the limit, short first and
    last batches, a pure function of its arguments (every feed uses it: same files, same fp64 sums)."""
    from tise_toolbox_amd.engine import item_schedule
    monkeypatch.delenv("TISE_RAMP", raising=False)
    assert item_schedule(30000, 50, 3000) == [50, 100, 250, 500, 1000, 1000, 1500, 2000, 2800, 2800] + [2750] * 6 + [1000, 500]
    assert item_schedule(12000, 50, 3000) == [50, 100, 250, 500, 1000, 1000, 1500, 2000, 2050, 2050, 1000, 500]
    assert item_schedule(3750, 50, 3000) == [50, 100, 250, 500, 1350, 1000, 500]            # a rank's share at 8 GPUs
    assert item_schedule(45, 5, 3000) == [45] and item_schedule(45, 5, 15) == [15, 15, 15] and item_schedule(0, 5, 15) == []
    assert item_schedule(500, 50, 3000) == [50, 100, 350] and item_schedule(80, 1, 3000) == [80]
    rng = np.random.default_rng(0)
    for _ in range(300):
        bs = int(rng.integers(1, 130))
        n = int(rng.integers(0, 700)) * bs
        limit = int(rng.integers(1, 80)) * bs
        s = item_schedule(n, bs, limit)
        assert sum(s) == n and all(x > 0 and x % bs == 0 and x <= limit for x in s), (n, bs, limit, s)
    monkeypatch.setenv("TISE_RAMP", "0")
    assert item_schedule(30000, 50, 3000) == [3000] * 10


def test_ipc_default_is_set_before_any_gpu_call(monkeypatch):
    """ROCr reads HSA_ENABLE_IPC_MODE_LEGACY once, when HIP is initialised -- torch.cuda.is_available() and
    set_device already do that -- so dist.init_from_env must have put the default into the environment BEFORE its first
    torch.cuda call (multi-rank, backend not gloo); an explicit value wins; gloo and one-rank groups leave it alone."""
    import torch
    from tise_toolbox_amd import dist as tdist
    seen = {}

    def spy(name, ret):
        def f(*a, **k):
            seen.setdefault(name, os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
            return ret
        return f
    monkeypatch.setattr(torch.cuda, "is_available", spy("is_available", True))
    monkeypatch.setattr(torch.cuda, "set_device", spy("set_device", None))
    monkeypatch.setattr(tdist.dist, "init_process_group", spy("init_process_group", None))
    monkeypatch.setattr(tdist.dist, "is_initialized", lambda: False)
    monkeypatch.setattr(tdist, "_FORCED", False)                           # restored afterwards (the forced call below sets it)
    for k, v in dict(RANK="1", WORLD_SIZE="4", LOCAL_RANK="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999").items():
        monkeypatch.setenv(k, v)
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("TISE_DIST_BACKEND", raising=False)
    assert tdist.init_from_env() == (1, 4, 1)
    assert seen == {"is_available": "0", "set_device": "0", "init_process_group": "0"}
    seen.clear()
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")                  # the user's explicit choice is kept
    tdist.init_from_env()
    assert seen["is_available"] == "1"
    seen.clear()
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY")
    tdist.init_from_env(backend="gloo")
    assert seen == {"init_process_group": None}
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    tdist.init_from_env(force=True)
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") is None
    assert tdist.collective_timeout().total_seconds() == 120.0
    monkeypatch.setenv("TISE_DIST_TIMEOUT_S", "7.5")
    assert tdist.collective_timeout().total_seconds() == 7.5


@pytest.mark.timeout(300)
def test_png_ring_native_workers_hand_unsupported_files_to_pillow(tmp_path, monkeypatch):
    """Round 6: the feed's first-line workers are the native program (csrc/png_worker.c).  Files outside its subset -- palette,
    gray, 16-bit, interlaced PNGs, a JPEG -- make it hand the chunk back; Python fallback workers (started on demand) redo
    those chunks with Pillow.  Pixels == Image.open(f).convert("RGB") for every file either way, in order; a directory with
    no such file never starts a Python process; TISE_PNG_WORKER=python keeps the round-5 workers."""
    from PIL import Image
    from tests import _cases
    from tise_toolbox_amd import png_ring
    imgs = _cases.smooth_images(37, 40, 56, seed=21)
    files = []
    for i, im in enumerate(imgs):
        p = tmp_path / f"{i:04d}.png"
        pil = Image.fromarray(im)
        if i % 9 == 2:
            pil.convert("P").save(p)
        elif i % 9 == 4:
            pil.convert("L").save(p)
        elif i == 15:
            pil.convert("RGBA").save(p)
        elif i == 20:
            p = tmp_path / f"{i:04d}png.jpg"
            pil.save(p, "JPEG")
        else:
            pil.save(p)
        files.append(str(p))
    want = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    assert png_ring.native_worker() is not None, "tise_png_worker was not built"
    for env, native in (({}, True), ({"TISE_PNG_WORKER": "python"}, False)):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ld = png_ring.PngRingLoader(files, 1, "cpu", workers=3, chunk=4)
        assert (ld.native is not None) == native
        got = np.zeros_like(want)
        for lo, view in ld.iter_host():
            got[lo:lo + len(view)] = view
            started_py = len(ld.py_procs)
        assert np.array_equal(got, want)
        assert (started_py > 0) == native                                 # the native workers needed Pillow's help; the Python ones are complete
    monkeypatch.delenv("TISE_PNG_WORKER")
    plain = [f for i, f in enumerate(files) if i % 9 not in (2, 4) and i != 20]
    ld = png_ring.PngRingLoader(plain, 1, "cpu", workers=2, chunk=3)
    n = 0
    for lo, view in ld.iter_host():
        n += len(view)
        assert not ld.py_procs
    assert n == len(plain)


def test_clip_preprocess_geometry_follows_torchvision_integer_rules():
    """clip._transform = torchvision Resize(224) + CenterCrop(224).  torchvision 0.9.1
    (transforms/functional.py resize: ``oh = int(size * h / w)`` for the longer side -- truncation; center_crop:
    ``int(round((h - th) / 2.))``) -- known answers worked by hand from those two lines."""
    from tise_toolbox_amd.clip_model import preprocess_geometry as g
    assert g(256, 256) == (224, 224, 0, 0)
    assert g(480, 640) == (224, 298, 0, 37)          # 224 * 640 / 480 = 298.67 -> 298 (round() would give 299); (298 - 224) / 2 = 37
    assert g(640, 480) == (298, 224, 37, 0)
    assert g(227, 300) == (224, 296, 0, 36)          # 296.04 -> 296; 72 / 2 = 36
    assert g(300, 225) == (298, 224, 37, 0)          # 298.67 -> 298
    assert g(229, 224) == (229, 224, 2, 0)           # (229 - 224) / 2 = 2.5 -> 2 (halves to even)
    assert g(231, 224) == (231, 224, 4, 0)           # 3.5 -> 4


def test_range_guard_hit_reruns_on_the_exact_path(monkeypatch, capsys):
    """a split-fp16 run whose range guard fires is finished in-process on the exact-fp32 path
    (engine.run_with_exact_fallback): second call with TISE_CONV=miopen; an exact run that raises is not retried."""
    from tise_toolbox_amd.engine import run_with_exact_fallback
    monkeypatch.setenv("TISE_CONV", "split")            # (registers the variable's ABSENCE for the teardown: the function sets it itself)
    calls = []

    def job():
        calls.append(os.environ.get("TISE_CONV", "split"))
        if calls[-1] == "split":
            raise FloatingPointError("range guard")
        return 42
    assert run_with_exact_fallback(job, "the test job") == 42 and calls == ["split", "miopen"]
    assert "again on the exact-fp32 convolution path" in capsys.readouterr().err
    monkeypatch.setenv("TISE_CONV", "miopen")

    def bad():
        raise FloatingPointError("still bad")
    with pytest.raises(FloatingPointError):
        run_with_exact_fallback(bad)


def test_ring_device_batches_are_clamped_by_the_real_image_size():
    """the callers size ``group`` for 256 x 256 images; the loader's device batches (and its three buffers) follow the REAL
    size: at most engine.STAGING_BYTES_CAP of pixels per buffer, whole loader batches, never less than one."""
    from tise_toolbox_amd import png_ring
    from tise_toolbox_amd.engine import STAGING_BYTES_CAP
    files = [f"{i:05d}.png" for i in range(1000)]
    ld = png_ring.PngRingLoader(files, 50, "cpu", group=60, start=False)
    ld.h = ld.w = 256
    assert sum(ld.item_sizes()) == 1000 and max(ld.item_sizes()) <= 3000
    for hw, want_max in ((1024, 300), (2048, 50), (8192, 50)):
        ld.h = ld.w = hw
        sizes = ld.item_sizes()
        assert sum(sizes) == 1000 and all(s % 50 == 0 for s in sizes)
        assert max(sizes) <= want_max and (max(sizes) * hw * hw * 3 <= STAGING_BYTES_CAP or max(sizes) == 50), (hw, sizes)


class _Tattler:
    """finaliser that reports the pid it ran in"""
    def __init__(self, path):
        self.path, self.me = path, self
    def __del__(self):
        with open(self.path, "a") as f:
            f.write(f"{os.getpid()}\n")


class _Churn(torch.utils.data.Dataset):
    """every item allocates enough container objects to push the worker's collector through all generations"""
    def __len__(self):
        return 8
    def __getitem__(self, i):
        import gc
        junk = [[j] for j in range(20000)]
        del junk
        gc.collect()
        return torch.tensor([os.getpid()])


def test_dataloader_workers_do_not_finalise_the_parents_garbage(tmp_path):
    """img_data.worker_init (gc.freeze in every forked DataLoader worker): a cycle that is garbage in the parent at fork time --
    in the product an engine with its HIP handles -- must not be finalised inside a worker; device.StatsAccumulator /
    FrechetSolver additionally refuse to destroy their handle from another pid."""
    import gc
    from tise_toolbox_amd import device, img_data
    log = tmp_path / "finalised.txt"
    gc.collect()
    gc.disable()
    try:
        _Tattler(str(log))                                         # unreachable at once, kept alive by its own cycle
        loader = torch.utils.data.DataLoader(_Churn(), batch_size=2, num_workers=2, worker_init_fn=img_data.worker_init)
        pids = set(int(p) for b in loader for p in b.flatten().tolist())
        del loader
    finally:
        gc.enable()
    assert os.getpid() not in pids and len(pids) >= 1               # the items really came from forked workers
    ran_in = set(int(x) for x in log.read_text().split()) if log.exists() else set()
    assert not (ran_in & pids), (ran_in, pids)                      # never inside a worker
    gc.collect()
    assert os.getpid() in set(int(x) for x in log.read_text().split())   # the parent collects it
    # the handle classes: close() from a foreign pid drops the handle without calling the library
    for cls in (device.StatsAccumulator, device.FrechetSolver):
        obj = cls.__new__(cls)
        obj._h, obj._pid = __import__("ctypes").c_void_p(1234), os.getpid() + 1
        obj.close()                                                # would crash in tise_*_destroy(0x4d2) if it called it
        assert not obj._h
