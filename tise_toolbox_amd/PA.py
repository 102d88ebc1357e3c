#!/usr/bin/env python3
"""Positional alignment: drop-in for positional_alignment/PA.py.

Same CLI (PA.py:17-24: --image_dir --pa_input_file --saved_file_path --gpu_id), same input pickle (dict phrase ->
list of {caption_id, caption, false_caption}), images at ``image_dir/<phrase>/<caption_id>.png`` (:56), same success
rule (softmax over [true, false] caption, entry 0 > 0.6, :37-42), same per-phrase score and mean (:52-67) and the
same result text ``PA = {value}`` (:70-74).  As in RP_coco.py the order of work changes: every distinct caption is
embedded once, the images in batches, and csrc/retrieval.hip scores all items of a phrase in one launch (p0 rounded
as the fp16 CLIP.forward rounds it).  Under torchrun the items of every phrase are sharded and the per-phrase
{success, total} pairs all-reduced.  Unlike the reference nothing runs at import.
"""
import argparse
import os
import pickle

import numpy as np
import torch

from . import _lib, clip_model, device, dist as tdist, weights as tweights
from .RP_coco import build_towers, embed_texts

THRESHOLD = 0.6                                                              # PA.py:41


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--image_dir", default="", type=str, help="Path to the folder containing generated images.")
    parser.add_argument("--pa_input_file", default="captions/PA_input_captions.pkl", type=str)
    parser.add_argument("--saved_file_path", default=None, type=str, help="Path to file saving result")
    parser.add_argument("--gpu_id", default="0", type=str)
    parser.add_argument("--weights", default=None, type=str, help="OpenAI CLIP ViT-B/32 state_dict (.pt); default: ~/.cache/clip/ViT-B-32.pt")
    parser.add_argument("--vocab", default=None, type=str, help="bpe_simple_vocab_16e6.txt.gz (required with real weights)")
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="seeded stand-in towers + word-hash tokenizer (plumbing / throughput only; results are tagged)")
    parser.add_argument("--batch-size", default=256, type=int)
    parser.add_argument("--num-workers", default=0, type=int, help="image decode processes (0 = auto)")
    parser.add_argument("--png-feed", default="ring", choices=["ring", "dataloader"])
    return parser.parse_args(argv)


def _embed_paths(model, paths, dev, batch, workers=0, feed="ring"):
    """PA.py:34: ``preprocess(Image.open(img_path))`` -- no conversion before clip's resize, so only plain RGB files take the
    ring + device preprocess (RP_coco.embed_paths)."""
    from .RP_coco import embed_paths
    return embed_paths(model, paths, dev, batch, workers, feed, convert_first=False)


def pa_successes(img_emb, txt_emb, pair_index, logit_scale):
    """(n,) 0/1: softmax([true, false])[0] > 0.6 (PA.py:37-42) for n items; pair_index (n, 2) int32 rows of txt_emb."""
    _, p0 = device.cosine_top1(img_emb, txt_emb, pair_index, normalize=False, logit_scale=logit_scale, want_p0=True)
    return (p0 > THRESHOLD).float()


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise _lib.TiseLibraryError("PA needs an MI355X: there is no CPU path")
    rank, world, local_rank = tdist.init_from_env()
    dev = torch.device(f"cuda:{local_rank}" if world > 1 else f"cuda:{args.gpu_id}")
    torch.cuda.set_device(dev)
    wpath, tag = tweights.resolve(args.weights, args.synthetic_weights, "clip")
    if wpath is not None and not args.vocab:
        raise RuntimeError("real CLIP weights need the BPE vocabulary: pass --vocab bpe_simple_vocab_16e6.txt.gz")
    model, scale = build_towers(wpath, dev)
    tokenizer = clip_model.BPETokenizer(args.vocab) if args.vocab else clip_model.HashTokenizer()
    with open(args.pa_input_file, "rb") as f:
        data = pickle.load(f)
    phrases = list(data.keys())                                               # PA.py:48
    sums = torch.zeros((len(phrases), 2), dtype=torch.float64, device=dev)    # {success, total} per phrase
    for pi, phrase in enumerate(phrases):
        items = data[phrase]
        lo, hi = tdist.shard_range(len(items), rank, world)
        mine = items[lo:hi]
        if not mine:
            continue
        table = {}
        index = np.asarray([[table.setdefault(it["caption"], len(table)), table.setdefault(it["false_caption"], len(table))]
                            for it in mine], dtype=np.int32)
        txt = embed_texts(model, tokenizer, list(table), dev, args.batch_size)
        img = _embed_paths(model, [os.path.join(args.image_dir, phrase, str(it["caption_id"]) + ".png") for it in mine],
                           dev, args.batch_size, args.num_workers, args.png_feed)
        ok = pa_successes(img, txt, torch.from_numpy(index).to(dev), scale)
        sums[pi, 0] = ok.double().sum()
        sums[pi, 1] = float(len(mine))
    tdist.all_reduce_sum_(sums)
    s = sums.cpu().numpy()
    # a phrase without items keeps the score 0.0 it was initialised with and still enters the mean (PA.py:52-67)
    phrase_res = {p: {"success": float(s[i, 0]), "total": float(s[i, 1]),
                      "score": float(s[i, 0]) / float(s[i, 1]) if s[i, 1] else 0.0}
                  for i, p in enumerate(phrases)}
    PA = np.mean([phrase_res[p]["score"] for p in phrase_res])                # :67
    if tdist.is_main():
        for p in phrases:
            print(p, phrase_res[p])                                           # :64
        if args.saved_file_path is not None:
            with open(args.saved_file_path, "w") as f:
                f.write(f"PA = {PA}{tag}")                                    # :70-71
        print(f"PA = {PA}{tag}")                                              # :74
    return PA


if __name__ == "__main__":
    tdist.run_cli(main)
