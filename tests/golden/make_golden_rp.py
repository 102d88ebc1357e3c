#!/usr/bin/env python3
"""Golden vectors for the R-precision / positional-alignment host logic.

The reference scripts (text_relevance/RP_coco.py, positional_alignment/PA.py) execute at import and need the
third-party `clip` package, its weights and its BPE vocabulary -- none of which exist here.  Everything AROUND the
CLIP towers (binning of the shuffled item ids, the `[true] + mismatched` candidate order, `argmax == 0`, bin
accuracy, mean / std, the result-file text; for PA the `softmax > 0.6` rule and the per-phrase mean) is the
reference's own code, so it is executed for real: a stub `clip` module is registered whose "model" returns
logits that are a fixed, seeded function of the image file and of the caption text; `random.seed` fixes the
shuffle.  The fixture stores the inputs (items, the logits the stub produced per item, the shuffle the seed
gives) and the text the reference wrote.  The real CLIP forward stays "parity unpinned".

    python tests/golden/make_golden_rp.py        (needs /root/reference; run in the build container)
"""
import hashlib
import json
import os
import pickle
import random
import runpy
import sys
import tempfile
import types

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
sys.dont_write_bytecode = True


def code(text):
    """Deterministic 16-d unit vector of a string."""
    h = hashlib.sha256(text.encode()).digest()
    v = np.frombuffer(h[:16], dtype=np.uint8).astype(np.float64) - 127.5
    return v / np.linalg.norm(v)


class StubModel:
    """logits_per_image = 100 * <image code, caption code>: image code comes from the pixel data."""
    def __init__(self, log):
        self.log = log

    def __call__(self, image, text):
        logits = torch.tensor(np.array([[100.0 * float(np.dot(image[0].numpy(), t.numpy())) for t in text]]))
        self.log.append(logits[0].numpy().copy())
        return logits, logits.t()


class OnHost:
    """What the scripts call on the stub's outputs: `.unsqueeze(0)` and `.to(device)` (PA.py asks for cuda)."""
    def __init__(self, t):
        self.t = t

    def unsqueeze(self, d):
        return OnHost(self.t.unsqueeze(d))

    def to(self, device):
        return self.t


def install_stub(log):
    clip = types.ModuleType("clip")
    captions_seen = []

    def load(name, device=None):
        assert name == "ViT-B/32"
        def preprocess(img):
            a = np.asarray(img.convert("RGB"), dtype=np.float64).reshape(-1)[:16] - 127.5
            return OnHost(torch.tensor(a / (np.linalg.norm(a) + 1e-12)))
        return StubModel(log), preprocess

    def tokenize(captions):
        captions_seen.append(list(captions))
        return OnHost(torch.tensor(np.stack([code(c) for c in captions])))

    clip.load, clip.tokenize = load, tokenize
    sys.modules["clip"] = clip
    return captions_seen


def write_image(path, rng, toward=None):
    a = rng.integers(0, 256, size=(4, 4, 3), dtype=np.uint8)
    if toward is not None:                                  # make the first 16 values resemble a caption's code
        v = np.clip(np.round(toward * 90 + 127.5 + rng.normal(0, 25, 16)), 0, 255).astype(np.uint8)
        a.reshape(-1)[:16] = v
    Image.fromarray(a).save(path)


def run_rp(n_items, n_mis, seed, out_name):
    rng = np.random.default_rng(seed)
    words = ["a", "red", "bus", "dog", "on", "the", "grass", "two", "people", "near", "table", "cat", "blue", "sky"]
    items = []
    with tempfile.TemporaryDirectory() as tmp:
        img_dir = os.path.join(tmp, "images"); os.makedirs(img_dir)
        for i in range(n_items):
            cap = " ".join(rng.choice(words, size=6)) + f" {i}"
            mis = [" ".join(rng.choice(words, size=6)) + f" m{i}_{j}" for j in range(n_mis)]
            cid = 1000 + 7 * i
            items.append({"caption_id": cid, "caption": cap, "mismatched_captions": mis})
            # about 60 % of the images are pulled toward their true caption
            write_image(os.path.join(img_dir, f"{cid}.png"), rng, code(cap) if rng.random() < 0.6 else None)
        pkl = os.path.join(tmp, "rp.pkl")
        pickle.dump(items, open(pkl, "wb"))
        saved = os.path.join(tmp, "out.txt")
        log = []
        install_stub(log)
        random.seed(seed)
        perm = list(range(n_items)); random.shuffle(perm)        # the shuffle the script is about to draw
        random.seed(seed)
        argv = sys.argv
        sys.argv = ["RP_coco.py", "--image_dir", img_dir, "--rp_input_file", pkl, "--saved_file_path", saved]
        try:
            runpy.run_path(os.path.join(REF, "text_relevance", "RP_coco.py"), run_name="__main__")
        finally:
            sys.argv = argv
        text = open(saved).read()
    # the stub logged logits in processing order = bins in order, items of a bin in order
    samples = n_items // 10
    order = []
    for b in range(10):
        order += perm[b * samples:] if (b == 9 and n_items % 10 != 0) else perm[b * samples:(b + 1) * samples]
    logits = np.zeros((n_items, 1 + n_mis))
    for k, item_idx in enumerate(order):
        logits[item_idx] = log[k]
    np.savez_compressed(os.path.join(HERE, out_name), logits=logits, perm=np.array(perm), seed=seed,
                        expected_text=text, n_items=n_items)
    print(out_name, text)


def run_pa(seed, out_name):
    rng = np.random.default_rng(seed)
    phrases = ["on top of", "under", "left of", "behind"]
    data, logit_rows = {}, {}
    with tempfile.TemporaryDirectory() as tmp:
        for p in phrases:
            os.makedirs(os.path.join(tmp, "images", p))
            data[p] = []
            for i in range(int(rng.integers(5, 12))):
                cap, false = f"a cup {p} a table {i}", f"a table {p} a cup {i}"
                cid = 50 + i
                data[p].append({"caption_id": cid, "caption": cap, "false_caption": false})
                write_image(os.path.join(tmp, "images", p, f"{cid}.png"), rng, code(cap) if rng.random() < 0.5 else None)
        pkl = os.path.join(tmp, "pa.pkl"); pickle.dump(data, open(pkl, "wb"))
        saved = os.path.join(tmp, "out.txt")
        log = []
        install_stub(log)
        argv = sys.argv
        sys.argv = ["PA.py", "--image_dir", os.path.join(tmp, "images"), "--pa_input_file", pkl, "--saved_file_path", saved]
        try:
            runpy.run_path(os.path.join(REF, "positional_alignment", "PA.py"), run_name="__main__")
        finally:
            sys.argv = argv
        text = open(saved).read()
    k = 0
    for p in phrases:
        logit_rows[p] = [log[k + i].tolist() for i in range(len(data[p]))]
        k += len(data[p])
    json.dump({"phrases": phrases, "logits": logit_rows, "expected_text": text},
              open(os.path.join(HERE, out_name), "w"), indent=1)
    print(out_name, text)


if __name__ == "__main__":
    torch.cuda.is_available = lambda: False                 # the scripts pick "cuda:<id>" otherwise
    run_rp(57, 9, 3, "rp_stub_57x10.npz")                   # 57 items: last bin takes the remainder (5 + 12)
    run_rp(40, 99, 5, "rp_stub_40x100.npz")                 # divisible: all bins equal, 100 candidates
    run_pa(7, "pa_stub.json")
