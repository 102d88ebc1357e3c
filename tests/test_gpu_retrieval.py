"""GPU parity of the R-precision / positional-alignment reduction (section 8 f3): csrc/retrieval.hip through the
C ABI against oracle/rp_oracle.py (fp64) and against what the reference scripts wrote (stub-CLIP fixtures)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from oracle import rp_oracle

pytestmark = pytest.mark.gpu


def _oracle_top1(img, txt, index, normalize, scale):
    n = img.shape[0]
    top1, p0, margin = np.zeros(n, np.int64), np.zeros(n), np.zeros(n)
    for i in range(n):
        cand = txt[index[i]] if index is not None else txt[i * (txt.shape[0] // n):(i + 1) * (txt.shape[0] // n)]
        lg = rp_oracle.clip_logits(img[i], cand, scale, normalize)
        top1[i] = int(np.argmax(lg))
        p0[i] = rp_oracle.softmax(lg)[0]
        s = np.sort(lg)
        margin[i] = s[-1] - s[-2] if len(s) > 1 else np.inf
    return top1, p0, margin


@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
@pytest.mark.parametrize("d,c,use_index,normalize", [(512, 100, True, True), (512, 100, True, False), (768, 7, False, True),
                                                     (100, 33, True, True), (1024, 2, False, False), (64, 1, True, True)])
def test_cosine_top1_matches_fp64_oracle(cuda_device, dtype, d, c, use_index, normalize):
    from tise_toolbox_amd import device
    rng = np.random.default_rng(d + c)
    n, rows = 1501, 4000
    img = rng.standard_normal((n, d)).astype(np.float32)
    if use_index:
        txt = rng.standard_normal((rows, d)).astype(np.float32)
        index = rng.integers(0, rows, size=(n, c)).astype(np.int32)
        # make about half of the items retrieve candidate 0
        for i in range(0, n, 2):
            txt[index[i, 0]] = img[i] + 0.7 * rng.standard_normal(d).astype(np.float32)
    else:
        txt = rng.standard_normal((n * c, d)).astype(np.float32)
        index = None
    if not normalize:
        img /= np.linalg.norm(img, axis=1, keepdims=True)
        txt /= np.linalg.norm(txt, axis=1, keepdims=True)
    ti = torch.from_numpy(img).to(cuda_device).to(dtype)
    tt = torch.from_numpy(txt).to(cuda_device).to(dtype)
    tidx = torch.from_numpy(index).to(cuda_device) if index is not None else None
    top1, p0 = device.cosine_top1(ti, tt, tidx, normalize=normalize, logit_scale=100.0)
    # the oracle sees exactly the values the kernel sees (the fp16 rounding is part of the input, not of the kernel)
    want1, wantp, margin = _oracle_top1(ti.float().cpu().numpy().astype(np.float64), tt.float().cpu().numpy().astype(np.float64),
                                        index, normalize, 100.0)
    got1 = top1.cpu().numpy()
    bad = got1 != want1
    assert not np.any(bad & (margin > 1e-4)), (int(bad.sum()), margin[bad][:5])     # index work: exact away from ties
    assert bad.sum() <= 2
    assert np.abs(p0.cpu().numpy() - wantp).max() <= 2e-5
    again, _ = device.cosine_top1(ti, tt, tidx, normalize=normalize, logit_scale=100.0)
    assert torch.equal(again, top1)
    if c > 1 and use_index:
        assert 0.2 < float((got1 == 0).mean()) < 0.9


@pytest.mark.parametrize("name", ["rp_stub_57x10.npz", "rp_stub_40x100.npz"])
def test_rp_through_the_kernel_reproduces_reference_text(cuda_device, golden_dir, name):
    """The fixture's logits (what the stub CLIP returned inside the reference script) are fed as 1-d 'embeddings'
    (txt_j = logit_j, img = 1, raw dot product): bins + kernel + reduction must write the reference's text."""
    from tise_toolbox_amd import RP_coco
    g = np.load(os.path.join(golden_dir, name))
    logits = g["logits"]
    n, c = logits.shape
    d = 64
    img = torch.zeros((n, d), device=cuda_device)
    img[:, 0] = 1.0
    txt = torch.zeros((n * c, d), device=cuda_device)
    txt[:, 0] = torch.from_numpy(logits.reshape(-1)).float().to(cuda_device)
    index = torch.arange(n * c, dtype=torch.int32, device=cuda_device).view(n, c)
    mean, std, _ = RP_coco.r_precision(img, txt, index, g["perm"].tolist(), normalize=False, logit_scale=1.0)
    assert f"R-precision: {mean} +- {std}" == str(g["expected_text"])


def test_pa_through_the_kernel_reproduces_reference_text(cuda_device, golden_dir):
    from tise_toolbox_amd import device
    g = json.load(open(os.path.join(golden_dir, "pa_stub.json")))
    scores = []
    for p in g["phrases"]:
        lg = torch.tensor(g["logits"][p], dtype=torch.float32, device=cuda_device)          # (n, 2) [true, false]
        n = lg.shape[0]
        img = torch.zeros((n, 64), device=cuda_device); img[:, 0] = 1.0
        txt = torch.zeros((2 * n, 64), device=cuda_device); txt[:, 0] = lg.reshape(-1)
        _, p0 = device.cosine_top1(img, txt, None, normalize=False, logit_scale=1.0)
        ok = (p0 > 0.6).float()                                                              # PA.py:41
        scores.append(float(ok.sum().item()) / n)
    assert f"PA = {np.mean(scores)}" == g["expected_text"]


def test_rp_cli_end_to_end_with_stand_in_towers(cuda_device, tmp_path):
    """Whole drop-in: pickle + PNGs -> towers (seeded stand-ins, fp16 like clip.load on a GPU) -> kernel -> file.
    Checked against the oracle applied to the SAME embeddings (the towers themselves: parity unpinned)."""
    from PIL import Image
    from tise_toolbox_amd import RP_coco, clip_model
    rng = np.random.default_rng(0)
    words = ["a", "red", "bus", "dog", "on", "the", "grass", "two", "people", "near", "table"]
    items = []
    img_dir = tmp_path / "images"
    img_dir.mkdir()
    pool = [" ".join(rng.choice(words, 5)) + f" x{k}" for k in range(30)]          # mismatched captions repeat
    for i in range(23):
        items.append({"caption_id": 100 + i, "caption": " ".join(rng.choice(words, 5)) + f" {i}",
                      "mismatched_captions": [pool[(i * 3 + 5 * j) % 30] for j in range(6)]})
        Image.fromarray(rng.integers(0, 256, (64, 80, 3), dtype=np.uint8)).save(img_dir / f"{100 + i}.png")
    pkl = tmp_path / "rp.pkl"
    pickle.dump(items, open(pkl, "wb"))
    out = tmp_path / "rp.txt"
    mean, std = RP_coco.main(["--image_dir", str(img_dir), "--rp_input_file", str(pkl), "--saved_file_path", str(out),
                              "--gpu_id", str(cuda_device.index or 0), "--seed", "4", "--batch-size", "8",
                              "--synthetic-weights"])
    from tise_toolbox_amd.weights import SYNTHETIC_TAG
    assert open(out).read() == f"R-precision: {mean} +- {std}" + SYNTHETIC_TAG
    # oracle on the same embeddings
    model = clip_model.build_clip().to(cuda_device).half()
    caps, index = RP_coco.caption_table(items)
    assert len(caps) < 23 * 7                                         # captions are de-duplicated
    txt = RP_coco.embed_texts(model, clip_model.HashTokenizer(), caps, cuda_device, 8).float().cpu().numpy().astype(np.float64)
    img = RP_coco.embed_images(model, str(img_dir), [it["caption_id"] for it in items], cuda_device, 8, workers=0)
    img = img.float().cpu().numpy().astype(np.float64)
    scale = float(model.logit_scale.exp())
    success, margins = [], []
    for i in range(len(items)):
        lg = rp_oracle.clip_logits(img[i], txt[index[i]], scale, normalize=False)
        success.append(int(np.argmax(lg) == 0))
        margins.append(np.sort(lg)[-1] - np.sort(lg)[-2])
    if min(margins) > 1e-3:
        m2, s2, _ = rp_oracle.rp_score(success, RP_coco.shuffled_ids(len(items), 4))
        assert (m2, s2) == (mean, std)
