#!/usr/bin/env python3
"""Where RP_coco.embed_texts spends its time (40 000 captions, batch 2048): tokeniser, H2D, tower, scatter."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd import RP_coco, clip_hip, clip_model

dev = torch.device("cuda:0")
model = clip_model.build_clip().to(dev).half()
towers = clip_hip.HipTowers(model)
caps = [f"a photo of item number {i} near the {['bus', 'dog', 'table', 'tree'][i % 4]}" for i in range(40000)]


def timed(label, fn, n=1):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        r = fn()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{label:58s} {dt * 1e3:8.1f} ms", flush=True)
    return r


tok = clip_model.HashTokenizer()
t_all = timed("tokenise 40 000 captions (cold memo)", lambda: tok(caps))
timed("tokenise 40 000 captions (warm memo)", lambda: tok(caps))
with torch.no_grad():
    for env in ("1", "0", "1", "0"):
        os.environ["TISE_CLIP_TRUNCATE"] = env
        timed(f"embed_texts, TISE_CLIP_TRUNCATE={env}", lambda: RP_coco.embed_texts(towers, clip_model.HashTokenizer(), caps, dev, 2048))
    t77 = t_all[:2048].to(dev)
    length = int(t_all[:2048].argmax(-1).max()) + 1
    t11 = t_all[:2048, :length].contiguous().to(dev)
    timed("encode_text 2048 x 77", lambda: towers.encode_text(t77), 5)
    timed(f"encode_text 2048 x {length}", lambda: towers.encode_text(t11), 5)
    f = towers.encode_text(t11)
    out = torch.empty((40000, 512), dtype=torch.float16, device=dev)
    sel = torch.arange(2048)
    timed("out[c0 + sel.to(dev)] = f", lambda: out.__setitem__(100 + sel.to(dev), f), 5)
    timed("t_all[sel] + .to(dev)", lambda: t_all[sel].to(dev), 5)
