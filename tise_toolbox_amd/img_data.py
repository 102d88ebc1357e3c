"""Image directory walking and decoding (mirror of image_realism/FID/img_data.py:7-35).

Same file-selection rule and order as the reference (``os.walk`` order, unsorted; a file is
kept when its NAME contains "jpg" or "png" anywhere, img_data.py:27-35 -- so ``.jpeg`` is
skipped and ``x.png.txt`` is taken, exactly as upstream).  What differs is what a sample is:
the reference resizes on the CPU worker (PIL) and ships fp32 CHW; here a sample is the
decoded uint8 HWC image and the resize runs on the GPU (``csrc/resize.hip``, bit-exact to
PIL), which cuts the worker->trainer traffic 5.5x (196 608 B vs 1 072 812 B per image).
"""
import os

import numpy as np
import torch
from PIL import Image
from torch.utils import data


def get_filenames(data_path):
    """img_data.py:27-35 / inception_score_star_coco.py:124-135, verbatim semantics."""
    images = []
    for path, subdirs, files in os.walk(data_path):
        for name in files:
            if name.rfind("jpg") != -1 or name.rfind("png") != -1:
                filename = os.path.join(path, name)
                if os.path.isfile(filename):
                    images.append(filename)
    return images


class Dataset(data.Dataset):
    """Characterizes a dataset for PyTorch (img_data.py:7).  ``transform`` is applied to the
    PIL image when given (reference behaviour); with ``transform=None`` the sample is the
    uint8 HWC tensor the device pipeline expects."""

    def __init__(self, path, transform=None, file_names=None):
        self.file_names = self.get_filenames(path) if file_names is None else list(file_names)
        self.transform = transform

    def __len__(self):
        return len(self.file_names)

    def __getitem__(self, index):
        img = Image.open(self.file_names[index]).convert("RGB")      # img_data.py:21
        if self.transform is not None:
            return self.transform(img)
        return torch.from_numpy(np.asarray(img).copy())              # (H, W, 3) uint8

    def get_filenames(self, data_path):
        return get_filenames(data_path)


def collate_u8(samples):
    """Stack equal-sized uint8 images to (B,H,W,3); otherwise keep a list (ragged crops, O-FID)."""
    if all(s.shape == samples[0].shape for s in samples):
        return torch.stack(samples, 0)
    return list(samples)
