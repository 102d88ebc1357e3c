#!/usr/bin/env python3
"""IS* (temperature-calibrated Inception Score) on MI355X.

Drop-in for the reference ``image_realism/IS/coco/inception_score_star_coco.py`` (functions
``get_inception_score`` :32, ``inception_score`` :138, ``load_data`` :124, ``preprocess`` :115; CLI
``--image_folder --saved_file --gpu`` :20-23; result text :153-156), with the reduction variants of
``image_realism/IS/bird/inception_score_star_bird.py:74-111,189-194`` and
``object_fidelity/O-IS/object_centric_inception_score.py:17-81`` selectable by flags.

Model note: the reference IS* runs the 2015 TensorFlow Inception graph (1008
classes, T calibrated for it); north_star prescribes the PyTorch InceptionV3 instead, so scores are
comparable between paths run with the SAME weights only.  What IS kept of the reference head: for
``--rule coco`` the logits are ``pool3 @ W.T`` WITHOUT the classifier bias, as
inception_score_star_coco.py:104-105 multiplies pool3 by the last layer's weight matrix only
(``--fc-bias on`` adds it; the bird and ois rules use the model's biased logits like their scripts).  The reduction itself -- temperature,
softmax, split rule, KL, exp, mean/std -- is the reference's, evaluated on device in fp64
(csrc/is_score.hip) from fp32 logits.
"""
import os
import sys
import warnings
from argparse import ArgumentDefaultsHelpFormatter, ArgumentParser

import numpy as np
import torch
import torch.utils.data

from . import _lib, device, dist as tdist, img_data, weights as tweights
from .engine import RealismEngine, T_BIRD, T_COCO, T_OIS, require_gpu

warnings.filterwarnings("ignore")

_ENGINE = None
_CONFIG = {"weights": None, "num_classes": 1000, "seed": 0, "temperature": T_COCO, "batch_size": 50,
           "rule": "coco", "drop_first_class": False, "num_workers": 0, "fc_bias": "auto", "png_feed": "ring"}


def configure(**kw):
    """Set weights / temperature / batch size used by the reference-signature functions below."""
    global _ENGINE
    _CONFIG.update(kw)
    _ENGINE = None


def _engine():
    global _ENGINE
    if _ENGINE is not None and getattr(_ENGINE, "_conv_mode", None) != os.environ.get("TISE_CONV", "split"):
        _ENGINE = None                                   # the convolution path changed (engine.run_with_exact_fallback): build anew
    if _ENGINE is None:
        _ENGINE = RealismEngine(dims=2048, weights=_CONFIG["weights"], num_classes=_CONFIG["num_classes"],
                                seed=_CONFIG["seed"], with_logits=True, fc_bias=_CONFIG["fc_bias"])
    _ENGINE._conv_mode = os.environ.get("TISE_CONV", "split")
    return _ENGINE


def inception_score_from_logits(logits, temperature=T_COCO, splits=10, rule="coco", drop_first_class=False,
                                idx_base=0, n_total=None, return_scores=False):
    """Reduction only: (N, C) fp32 logits (CUDA tensor or numpy) -> (mean, std).

    coco / bird: inception_score_star_coco.py:52-60 with tf.div(logits, T) + softmax (:107-108);
    ois: object_centric_inception_score.py:69-81 (splits of N // splits rows, tail dropped).
    """
    require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    if not isinstance(logits, torch.Tensor):
        logits = torch.as_tensor(np.ascontiguousarray(logits, dtype=np.float32))
    logits = logits.to(dev, torch.float32).contiguous()
    n = logits.shape[0] if n_total is None else n_total
    acc = device.InceptionScoreAccumulator(logits.shape[1], n, temperature, splits, rule, drop_first_class, dev)
    acc.update(logits, idx_base)
    tdist.all_reduce_sum_(acc.acc)
    mean, std, scores = acc.finalize()
    return (mean, std, scores) if return_scores else (mean, std)


def load_data(fullpath):
    """inception_score_star_coco.py:124-135: file names in os.walk order."""
    print("[Data] Read data from " + fullpath)
    images = img_data.get_filenames(fullpath)
    print("[Data] [{}] ...   ".format(len(images)))
    return images


def preprocess(img):
    """inception_score_star_coco.py:115-121 on a uint8 HWC array: gray -> 3 channels, bilinear 299x299,
    float32 in 0..255, batch axis.  Kept for API parity; the device path resizes in csrc/resize.hip."""
    from PIL import Image
    img = np.asarray(img)
    if len(img.shape) == 2:
        img = np.resize(img, (img.shape[0], img.shape[1], 3))
    img = np.asarray(Image.fromarray(img.astype(np.uint8)).resize((299, 299), Image.BILINEAR))
    return np.expand_dims(img.astype(np.float32), 0)


def get_inception_score(images, splits=10):
    """inception_score_star_coco.py:32-60: list of image file names -> (mean, std).

    Batched on the device instead of one ``sess.run`` per image (:34,:50); split membership is by
    global index in the given order (:55).  Under torchrun every rank takes a contiguous index range.
    """
    eng = _engine()
    n = len(images)
    bs = _CONFIG["batch_size"]
    rank, world, _ = tdist.env_world()
    lo, hi = tdist.shard_range(n, rank, world)
    # --batch-size is the loader's batch; a trunk pass takes up to engine.device_batch_images of them (split membership
    # is by global index, so batching changes nothing: tests/test_gpu_kernels.py batch invariance)
    from . import png_ring
    from .engine import coalesce_batches, device_batch_images
    workers = _CONFIG["num_workers"] if _CONFIG["num_workers"] and _CONFIG["num_workers"] > 0 else png_ring.auto_workers(world)

    def run(feed):
        eng.begin(n_total=n, temperature=_CONFIG["temperature"], splits=splits, rule=_CONFIG["rule"],
                  drop_first_class=_CONFIG["drop_first_class"])
        eng.reserve_activations(min(hi - lo, device_batch_images(bs)))     # one allocation of the passes' peak (engine.reserve_activations)
        base = lo
        for batch in feed:
            if isinstance(batch, (list, tuple)):              # images of different sizes: one trunk pass for the batch
                eng.step_u8_list(batch, base)
                base += len(batch)
            else:
                eng.step_u8(batch.to(eng.device, non_blocking=True), base)
                base += batch.shape[0]

    def dataloader_feed():
        dataset = img_data.Dataset(None, transform=None, file_names=images[lo:hi])
        loader = torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=False, drop_last=False,
                                             num_workers=min(32, workers), collate_fn=img_data.collate_u8, pin_memory=True,
                                             worker_init_fn=img_data.worker_init)
        return coalesce_batches(loader, eng.device, device_batch_images(bs))

    if _CONFIG.get("png_feed", "ring") == "ring" and hi > lo:
        # decode processes -> shared page-locked ring -> side-stream H2D (png_ring.py); every image is used (no drop-last:
        # inception_score_star_coco.py:44-51 feeds the images one by one), so the ring's loader batch is 1
        ring = png_ring.PngRingLoader(images[lo:hi], 1, eng.device, group=device_batch_images(1), workers=workers)
        try:
            run(ring)
        except png_ring.RaggedImages as e:
            if world > 1:
                raise RuntimeError(f"the ring feed under torchrun needs images of one size ({e})") from e
            print(f"[tise] png feed: images of different sizes ({e}); falling back to the DataLoader path", file=sys.stderr)
            run(dataloader_feed())
        finally:
            ring.close()
    else:
        run(dataloader_feed())
    eng.reduce()
    return eng.inception_score()


def inception_score(path):
    """inception_score_star_coco.py:138-141."""
    images = load_data(path)
    mean, std = get_inception_score(images)
    return mean, std


def _build_parser():
    parser = ArgumentParser(formatter_class=ArgumentDefaultsHelpFormatter)
    parser.add_argument("--image_folder", type=str, default="")
    parser.add_argument("--saved_file", type=str, default="")
    parser.add_argument("--gpu", type=int, default=0)
    parser.add_argument("--temperature", type=float, default=T_COCO,
                        help="IS* temperature (coco %r, bird %r, O-IS %r)" % (T_COCO, T_BIRD, T_OIS))
    parser.add_argument("--splits", type=int, default=10)
    parser.add_argument("--rule", type=str, default="coco", choices=["coco", "bird", "ois"])
    parser.add_argument("--drop-first-class", action="store_true", help="bird: class 0 is background")
    parser.add_argument("--batch-size", type=int, default=50)
    parser.add_argument("--weights", type=str, default=None, help="torchvision-format InceptionV3 state_dict (.pth)")
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="seeded stand-in parameters (plumbing / throughput only; results are tagged)")
    parser.add_argument("--num-classes", type=int, default=1000)
    parser.add_argument("--seed", type=int, default=0, help="seed of the --synthetic-weights parameters")
    parser.add_argument("--label", type=str, default="IS", choices=["IS", "O-IS", "bird"])
    parser.add_argument("--fc-bias", type=str, default="auto", choices=["auto", "on", "off"],
                        help="classifier bias in the logits. auto follows the reference script of --rule: coco forms its logits "
                             "from the weight matrix alone (inception_score_star_coco.py:104-105: no bias), bird and ois use "
                             "the model's biased logits")
    return parser


def main(argv=None):
    args = _build_parser().parse_args(argv)
    rank, world, _ = tdist.init_from_env()
    if world == 1:
        os.environ.setdefault("HIP_VISIBLE_DEVICES", str(args.gpu))   # :146
    wpath, tag = tweights.resolve(args.weights, args.synthetic_weights,
                                  "inception80" if args.label == "O-IS" and args.num_classes == 80 else "inception")
    configure(weights=wpath, num_classes=args.num_classes, seed=args.seed, temperature=args.temperature,
              batch_size=args.batch_size, rule=args.rule, drop_first_class=args.drop_first_class, fc_bias=args.fc_bias)
    images = load_data(args.image_folder)
    print(".......")
    from .engine import run_with_exact_fallback
    mean, std = run_with_exact_fallback(lambda: get_inception_score(images, splits=args.splits), "the Inception Score")
    if tdist.is_main():
        if args.label == "O-IS":                                       # object_centric_inception_score.py:126-129
            text = f"O-IS: {mean} +-  {std}"
            shown = f"O-IS: {mean} +- {std}"
        elif args.label == "bird":                                     # inception_score_star_bird.py:208-209
            text = shown = f"IS = {mean}  +-  {std}"
        else:                                                          # inception_score_star_coco.py:153-156
            text = "[Inception Score] mean: {:.5f} std: {:.5f}".format(mean, std)
            shown = "[Inception Score] mean: {:.2f} std: {:.2f}".format(mean, std)
        if args.saved_file:
            with open(args.saved_file, "w") as f:
                f.write(text + tag)
        print(shown + tag)
    return mean, std


if __name__ == "__main__":
    tdist.run_cli(main)
