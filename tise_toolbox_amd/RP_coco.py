#!/usr/bin/env python3
"""R-precision on COCO captions: drop-in for text_relevance/RP_coco.py.

Same CLI (RP_coco.py:17-25: --image_dir --rp_input_file --saved_file_path --gpu_id), same input pickle (list of
{caption_id, caption, mismatched_captions}), same bins (:41-52), same success rule (:72-78), same result text
(:85-90).  What changes is the order of work: the reference runs CLIP once per item with batch 1 on the image and
~100 captions (30 k items -> 3 M text-tower passes); here
    1. the DISTINCT captions of the whole input are tokenised and embedded once, in batches (text table),
    2. the images are embedded in batches,
    3. csrc/retrieval.hip scores all items at once against int32 index lists into the text table.
The shuffle that forms the bins is unseeded in the reference (:43); `--seed` makes it reproducible (default: unseeded
like the reference).  The towers are clip_model.py (see its header: scaffolding on library kernels, parity unpinned).

Data parallel (SURVEY 8e "RP"): under torchrun the ITEMS are sharded -- rank r embeds the images and the distinct
captions of its contiguous item range and scores them -- the bins come from one permutation every rank shares (rank
0's seed is broadcast when --seed is not given), and the only exchange is one all-reduce(SUM) of the per-bin
{success, count} pairs (160 bytes).
"""
import argparse
import os
import pickle
import random
import sys

import numpy as np
import torch

from . import _lib, clip_model, device, dist as tdist, img_data, weights as tweights


def parse_args(argv=None):
    parser = argparse.ArgumentParser(description="Calculating R-precision")
    parser.add_argument("--image_dir", default="", type=str, help="Path to the folder containing generated images.")
    parser.add_argument("--rp_input_file", default="captions/COCO_RP_captions.pkl", type=str)
    parser.add_argument("--saved_file_path", default=None, type=str, help="Path to file saving result")
    parser.add_argument("--gpu_id", default="0", type=str)
    parser.add_argument("--weights", default=None, type=str, help="OpenAI CLIP ViT-B/32 state_dict (.pt); default: ~/.cache/clip/ViT-B-32.pt")
    parser.add_argument("--vocab", default=None, type=str, help="bpe_simple_vocab_16e6.txt.gz (required with real weights)")
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="seeded stand-in towers + word-hash tokenizer (plumbing / throughput only; results are tagged)")
    parser.add_argument("--seed", default=None, type=int, help="seed of the bin shuffle (reference: unseeded)")
    parser.add_argument("--batch-size", default=256, type=int)
    parser.add_argument("--num-workers", default=0, type=int, help="image decode processes (0 = auto: the CPUs the process may really use)")
    parser.add_argument("--png-feed", default="ring", choices=["ring", "dataloader"],
                        help="ring: decode processes -> shared pinned ring, clip's preprocess on the device; dataloader: preprocess on DataLoader workers")
    return parser.parse_args(argv)


def build_towers(weights_path, dev):
    """-> (towers, logit_scale).  ``towers.encode_image`` / ``.encode_text`` in fp16, as the model clip.load serves on a GPU.
    Default: the hand-written kernels of csrc/clip_ops.hip (clip_hip.HipTowers: 1.4x / 2.4x the library-kernel module
    on the image / text tower, profiles/r02x_clip_towers.txt); TISE_CLIP=torch runs the module itself on PyTorch-ROCm
    library kernels (what north_star prescribes for the forward passes; same parameters, same results to fp16 rounding)."""
    model = clip_model.build_clip(weights_path)
    scale = float(model.logit_scale.detach().exp())            # clip's convert_weights leaves logit_scale in fp32
    model = model.to(dev).half()
    tdist.broadcast_module_(model)
    if os.environ.get("TISE_CLIP", "hip") == "torch":
        return model, scale
    from . import clip_hip
    return clip_hip.HipTowers(model, dev), scale


def make_bins(num_captions, perm, num_bins=10):
    """RP_coco.py:41-52."""
    samples_per_bin = int(len(perm) / num_bins)
    bins = []
    for i in range(num_bins):
        if i == (num_bins - 1) and num_captions % num_bins != 0:
            bins.append(perm[i * samples_per_bin:])
        else:
            bins.append(perm[i * samples_per_bin:(i + 1) * samples_per_bin])
    return bins


def shuffled_ids(num_captions, seed=None):
    ids = list(range(num_captions))
    (random.Random(seed) if seed is not None else random).shuffle(ids)
    return ids


def caption_table(rp_input):
    """Distinct captions and the (N, 1 + mismatched) int32 index lists, candidate 0 = the true caption (:68-70).
    Items must carry the same number of mismatched captions (they do in COCO_RP_captions.pkl: 99)."""
    table, index = {}, []
    for item in rp_input:
        row = [table.setdefault(c, len(table)) for c in [item["caption"]] + list(item["mismatched_captions"])]
        index.append(row)
    widths = {len(r) for r in index}
    if len(widths) != 1:
        raise ValueError(f"items carry different numbers of candidate captions: {sorted(widths)}")
    return list(table), np.asarray(index, dtype=np.int32)


def r_precision_from_success(success, perm, num_bins=10):
    """success: (N,) 0/1 per item.  Returns (mean, std, bin scores) as RP_coco.py:79-84."""
    success = np.asarray(success)
    scores = []
    for b in make_bins(len(success), perm, num_bins):
        scores.append(int(success[np.asarray(b, dtype=np.int64)].sum()) * 1.0 / len(b))
    return np.mean(scores), np.std(scores), scores


def bin_sums(success, item_base, perm, num_bins=10):
    """Per-bin [successes, count] (num_bins, 2) float64 of the items item_base .. item_base + len(success) - 1 under
    the GLOBAL permutation `perm`: additive over item shards; scores = sums[:, 0] / sums[:, 1] (RP_coco.py:79)."""
    bin_of = np.empty(len(perm), dtype=np.int64)
    for b, ids in enumerate(make_bins(len(perm), perm, num_bins)):
        bin_of[np.asarray(ids, dtype=np.int64)] = b
    sums = np.zeros((num_bins, 2), dtype=np.float64)
    mine = bin_of[item_base:item_base + len(success)]
    np.add.at(sums[:, 0], mine, np.asarray(success, dtype=np.float64))
    np.add.at(sums[:, 1], mine, 1.0)
    return sums


def r_precision_from_bin_sums(sums):
    scores = [int(s) * 1.0 / int(c) for s, c in sums]               # success_count * 1.0 / len(b)   (:79)
    return np.mean(scores), np.std(scores), scores


def r_precision(img_emb, txt_emb, txt_index, perm, normalize=True, logit_scale=100.0, num_bins=10):
    """Embeddings on the GPU -> (mean, std, bin scores); the scoring runs in csrc/retrieval.hip."""
    top1, _ = device.cosine_top1(img_emb, txt_emb, txt_index, normalize=normalize, logit_scale=logit_scale, want_p0=False)
    return r_precision_from_success((top1 == 0).cpu().numpy().astype(np.int64), perm, num_bins)


@torch.no_grad()
def embed_texts(model, tokenizer, captions, dev, batch):
    """Normalised text embeddings, in order.  Tokenising (host, one core) and encoding (device, asynchronous) overlap
    chunk by chunk; tokenising in DataLoader worker processes was tried and lost -- forking eight workers from a process
    that holds a GPU context took 7 s for a 0.7 s job.

    Round 4, hand-written towers only: the text transformer is CAUSAL (clip model.py build_attention_mask) and the feature
    is taken at the end-of-text token (CLIP.encode_text: ``x[arange, text.argmax(-1)]``), so nothing behind that token can
    reach the result -- the padding of a 77-token context is dead work.  Captions are therefore sorted by length inside a
    chunk of 4 batches and every batch is encoded at ITS longest caption (COCO captions: ~12-20 tokens instead of 77),
    each row's arithmetic being what it is at 77 tokens (per-token LayerNorm / GEMM rows, the same keys per query)."""
    out = torch.empty((len(captions), 0), dtype=torch.float16, device=dev)
    truncate = hasattr(model, "blocks_t") and os.environ.get("TISE_CLIP_TRUNCATE", "1") != "0"     # clip_hip.HipTowers
    chunk = 4 * batch if truncate else batch
    for c0 in range(0, len(captions), chunk):
        # index work in numpy: torch's CPU operators fan small tensors out over every core of the host (a 256-thread
        # OpenMP team per argsort / gather cost more than the text tower at 11 tokens)
        tok = tokenizer(captions[c0:c0 + chunk]).numpy()
        if truncate:
            length = tok.argmax(-1) + 1                               # position of the end-of-text token (the largest id) + 1
            order = np.argsort(length, kind="stable")
        else:
            order = np.arange(tok.shape[0])
        for i in range(0, tok.shape[0], batch):
            sel = order[i:i + batch]
            t = tok[sel, :max(2, int(length[sel].max()))] if truncate else tok[sel]
            f = model.encode_text(torch.from_numpy(np.ascontiguousarray(t)).to(dev))
            f = f / f.norm(dim=-1, keepdim=True)
            if out.shape[1] == 0:
                out = torch.empty((len(captions), f.shape[1]), dtype=f.dtype, device=dev)
            if truncate:
                out[torch.from_numpy(c0 + sel).to(dev)] = f
            else:
                out[c0 + i:c0 + i + f.shape[0]] = f
    return out.contiguous()


@torch.no_grad()
def embed_paths(model, paths, dev, batch, workers=0, feed="ring", convert_first=True):
    """Normalised image embeddings of the files ``paths``, in order.  Round 5: the images come through the PNG ring
    (png_ring.py: decode processes -> shared page-locked ring -> side-stream H2D) as uint8 and clip's preprocess runs on the
    device (clip_model.preprocess_device: Pillow-exact bicubic resample + crop + ToTensor / Normalize table) -- the reference's
    ``preprocess(Image.open(f))`` per item (RP_coco.py:64, PA.py:34) on eight DataLoader workers shipped 602 KB of fp32 per image
    through worker queues and bounded the CLI at ~3 k images/s.  ``convert_first``: RP_coco.py:64 converts to RGB BEFORE the
    preprocess (``preprocess(Image.open(p).convert("RGB"))``: what the ring's workers do); PA.py:34 does not
    (``preprocess(Image.open(p))``: clip resizes first and Pillow resamples RGBA premultiplied), so there only plain RGB files
    may take the ring.  Files of different sizes (or, for PA, not plain RGB) take the DataLoader path."""
    from . import png_ring
    out = []

    def consume(x):
        f = model.encode_image(x.half())
        out.append(f / f.norm(dim=-1, keepdim=True))
    workers = int(workers) if workers and int(workers) > 0 else png_ring.auto_workers(tdist.world_size())
    if feed == "ring" and len(paths):
        ring = png_ring.PngRingLoader(paths, 1, dev, group=batch, workers=workers, rgb_only=not convert_first)
        try:
            for u8 in ring:
                consume(clip_model.preprocess_device(u8))
            return torch.cat(out).contiguous()
        except png_ring.RaggedImages as e:
            print(f"[tise] png feed: {e}; falling back to the DataLoader path", file=sys.stderr)
            out.clear()
        finally:
            ring.close()
    loader = torch.utils.data.DataLoader(_Paths(paths, convert_first), batch_size=batch, shuffle=False, num_workers=min(32, workers),
                                         worker_init_fn=img_data.worker_init)
    for x in loader:
        consume(x.to(dev))
    return torch.cat(out).contiguous() if out else torch.empty((0, 512), dtype=torch.float16, device=dev)


class _Paths(torch.utils.data.Dataset):
    def __init__(self, paths, convert_first=True):
        self.paths, self.convert_first = paths, convert_first

    def __len__(self):
        return len(self.paths)

    def __getitem__(self, i):
        from PIL import Image
        img = Image.open(self.paths[i])
        return clip_model.preprocess(img.convert("RGB") if self.convert_first else img)   # (clip's preprocess converts to RGB itself, after the resize)


def embed_images(model, image_dir, caption_ids, dev, batch, workers=0, feed="ring"):
    return embed_paths(model, [os.path.join(image_dir, str(i) + ".png") for i in caption_ids], dev, batch, workers, feed)


def main(argv=None):
    args = parse_args(argv)
    if not torch.cuda.is_available():
        raise _lib.TiseLibraryError("RP_coco needs an MI355X: there is no CPU path")
    rank, world, local_rank = tdist.init_from_env()
    dev = torch.device(f"cuda:{local_rank}" if world > 1 else f"cuda:{args.gpu_id}")
    torch.cuda.set_device(dev)
    wpath, tag = tweights.resolve(args.weights, args.synthetic_weights, "clip")
    if wpath is not None and not args.vocab:
        raise RuntimeError("real CLIP weights need the BPE vocabulary: pass --vocab bpe_simple_vocab_16e6.txt.gz")
    import time
    timing = os.environ.get("TISE_TIMING") == "1" and rank == 0
    t_ph = [time.perf_counter()]

    def phase(label):
        if timing:
            torch.cuda.synchronize()
            t_ph.append(time.perf_counter())
            print(f"[tise timing] {label}: {t_ph[-1] - t_ph[-2]:.2f} s", file=sys.stderr, flush=True)
    model, scale = build_towers(wpath, dev)                            # clip.load on a GPU serves fp16 weights
    tokenizer = clip_model.BPETokenizer(args.vocab) if args.vocab else clip_model.HashTokenizer()
    phase("towers built")
    with open(args.rp_input_file, "rb") as f:
        rp_input = pickle.load(f)
    phase("pickle loaded")
    n_items = len(rp_input)
    seed = args.seed
    if world > 1:                                           # one permutation for all ranks
        t = torch.tensor([seed if seed is not None else random.randrange(2 ** 31)], dtype=torch.int64, device=dev)
        torch.distributed.broadcast(t, src=0)
        seed = int(t.item())
    perm = shuffled_ids(n_items, seed)
    lo, hi = tdist.shard_range(n_items, rank, world)
    mine = rp_input[lo:hi]
    sums = np.zeros((10, 2), dtype=np.float64)
    if mine:
        captions, index = caption_table(mine)
        phase("caption table")
        txt = embed_texts(model, tokenizer, captions, dev, args.batch_size)
        phase("text tower")
        img = embed_images(model, args.image_dir, [it["caption_id"] for it in mine], dev, args.batch_size, args.num_workers, args.png_feed)
        phase("images")
        # features are already normalised in the model's dtype, as CLIP.forward does before the matmul
        top1, _ = device.cosine_top1(img, txt, torch.from_numpy(index).to(dev), normalize=False, logit_scale=scale, want_p0=False)
        sums = bin_sums((top1 == 0).cpu().numpy(), lo, perm)
        phase("retrieval + bins")
    acc = torch.from_numpy(sums).to(dev)
    tdist.all_reduce_sum_(acc)                              # per-bin {success, count}: the only exchange
    mean, std, scores = r_precision_from_bin_sums(acc.cpu().numpy())
    if tdist.is_main():
        for bin_idx, s in enumerate(scores):
            print(f"Bin: {bin_idx}, RP: {s}")
        print(f"R-precision: {mean} +- {std}{tag}")
        if args.saved_file_path is not None:
            with open(args.saved_file_path, "w") as f:
                f.write(f"R-precision: {mean} +- {std}{tag}")
    return mean, std


if __name__ == "__main__":
    tdist.run_cli(main)
