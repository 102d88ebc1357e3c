#!/usr/bin/env python3
"""O-IS (object-centric Inception Score) on MI355X -- drop-in for the reference
``object_fidelity/O-IS/object_centric_inception_score.py``.

Mirrors ``inception_score(imgs, cuda=True, batch_size=32, resize=False, splits=1)`` (:17-81),
``IgnoreLabelDataset`` (:84-104: ``os.listdir`` order, ``convert("RGB")``, Resize((299, 299)) + ToTensor +
Normalize(0.5, 0.5)) and the CLI ``--image_dir --saved_file --gpu_id`` with result text
``O-IS: {mean} +-  {std}`` (:107-129).  Unlike the reference nothing runs at import.

Model: InceptionV3 fine-tuned to 80 COCO classes (``fc = Linear(2048, 80)``, ``transform_input=False``,
weights ``weights/inceptionv3_fine_to_with_80_coco_classes.pth``, :41-45); logits / 2.1737587451934814,
softmax (:53-56); per split ``exp(mean_i KL(p(y|x_i) || p(y)))`` over ``N // splits`` rows, the tail dropped
(:72-79) -- the ``ois`` rule of ``csrc/is_score.hip``, evaluated in fp64 on the device.
"""
import argparse
import os

import numpy as np
import torch
import torch.utils.data
from PIL import Image

from . import device, dist as tdist, img_data, weights as tweights
from .engine import RealismEngine, T_OIS, require_gpu
from .inception import InceptionV3

DEFAULT_WEIGHTS = "weights/inceptionv3_fine_to_with_80_coco_classes.pth"   # object_centric_inception_score.py:45


class IgnoreLabelDataset(torch.utils.data.Dataset):
    """object_centric_inception_score.py:84-104.  A sample is the decoded uint8 HWC image; the resize to
    299x299, ToTensor and Normalize((.5,.5,.5),(.5,.5,.5)) run on the device (PIL-exact resize kernel)."""

    def __init__(self, imgspath):
        self.imgspath = imgspath
        self.namelist = os.listdir(self.imgspath)                            # :93 (listdir, not walk)

    def __getitem__(self, index):
        img = Image.open(os.path.join(self.imgspath, self.namelist[index])).convert("RGB")   # :96-99
        return torch.from_numpy(np.asarray(img).copy())

    def __len__(self):
        return len(self.namelist)


_ENGINES = {}


def _engine(weights, num_classes, seed):
    key = (weights, num_classes, seed, os.environ.get("TISE_CONV", "split"))   # the convolution path is part of the key (exact-path rerun)
    if key not in _ENGINES:
        model = InceptionV3([3], normalize_input=False, weights=weights, num_classes=num_classes, seed=seed,
                            calibration="pm1")
        _ENGINES[key] = RealismEngine(dims=2048, model=model, with_logits=True,
                                      lut=device.make_lut(normalize_input=False, scale_pm1=True))
    return _ENGINES[key]


def inception_score(imgs, cuda=True, batch_size=32, resize=False, splits=1, weights=None, num_classes=80, seed=0,
                    temperature=T_OIS, num_workers=8):
    """Computes the inception score of the generated images imgs (object_centric_inception_score.py:17).

    imgs -- dataset of uint8 (H,W,3) images (``IgnoreLabelDataset``) or of (3,H,W) float tensors already
            normalised to [-1, 1] (the reference convention; ``resize`` then upsamples them to 299x299, :49)
    Returns (mean, std) over the splits, np.float64 like the reference.
    """
    N = len(imgs)
    assert batch_size > 0                                                     # :25
    assert N > batch_size                                                     # :26
    if not cuda:
        require_gpu()
        raise RuntimeError("cuda=False: tise_toolbox_amd has no CPU path")
    eng = _engine(weights, num_classes, seed)
    rank, world, _ = tdist.env_world()
    lo, hi = tdist.shard_range(N, rank, world)
    subset = torch.utils.data.Subset(imgs, range(lo, hi))
    loader = torch.utils.data.DataLoader(subset, batch_size=batch_size, num_workers=num_workers,
                                         collate_fn=img_data.collate_u8 if isinstance(imgs, IgnoreLabelDataset) else None,
                                         worker_init_fn=img_data.worker_init)
    eng.begin(n_total=N, temperature=temperature, splits=splits, rule="ois")
    base = lo
    # batch_size (32 in the reference's call, :122) is the loader's batch; a trunk pass takes up to
    # engine.device_batch_images of them
    from .engine import coalesce_batches, device_batch_images
    for batch in coalesce_batches(loader, eng.device, device_batch_images(batch_size)):
        if isinstance(batch, (list, tuple)):              # crops of different sizes: resized into ONE batch
            feats, logits = eng.features_from_u8_list(batch)
        elif batch.dtype == torch.uint8:
            feats, logits = eng.features_from_u8(batch.to(eng.device, non_blocking=True))
        else:
            x = batch.to(eng.device).float()
            if resize:
                x = torch.nn.functional.interpolate(x, size=(299, 299), mode="bilinear")         # :38,49
            feats, logits = eng._trunk(x.contiguous(memory_format=torch.channels_last), prenormalized=True)
        eng.is_acc.update(logits, base)
        base += logits.shape[0]
    eng.check_numerics()                                                      # split-fp16 range guard
    tdist.all_reduce_sum_(eng.is_acc.acc)
    mean, std, _ = eng.is_acc.finalize()
    return np.float64(mean), np.float64(std)


def parse_args(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--image_dir", default="", type=str)
    parser.add_argument("--saved_file", default="", type=str)
    parser.add_argument("--gpu_id", default=0, type=int)
    parser.add_argument("--weights", default=None, type=str, help=f"reference default: {DEFAULT_WEIGHTS}")
    parser.add_argument("--synthetic-weights", action="store_true",
                        help="seeded stand-in parameters (plumbing / throughput only; results are tagged)")
    parser.add_argument("--seed", default=0, type=int, help="seed of the --synthetic-weights parameters")
    return parser.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    rank, world, _ = tdist.init_from_env()
    if world == 1:
        torch.cuda.set_device(args.gpu_id)                                    # :121
    wpath, tag = tweights.resolve(args.weights, args.synthetic_weights, "inception80")       # :45
    print("Load images from: ", args.image_dir)
    imgs = IgnoreLabelDataset(args.image_dir)
    print("Calculating Inception Score...")
    from .engine import run_with_exact_fallback
    IS_mean, IS_std = run_with_exact_fallback(lambda: inception_score(imgs, cuda=True, batch_size=32, resize=False, splits=10,      # :122
                                                                      weights=wpath, seed=args.seed), "the O-IS")
    if tdist.is_main():
        if args.saved_file:
            with open(args.saved_file, "w") as f:
                f.write(f"O-IS: {IS_mean} +-  {IS_std}{tag}")                # :126-127
        print(f"O-IS: {IS_mean} +- {IS_std}{tag}")                           # :129
    return IS_mean, IS_std


if __name__ == "__main__":
    tdist.run_cli(main)
