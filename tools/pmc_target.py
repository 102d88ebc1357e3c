#!/usr/bin/env python3
"""Small target for rocprofv3 --pmc passes: the hand-written kernels at bench sizes, no MIOpen.
  rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -o fetch -- python3 tools/pmc_target.py
  rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d out -o write -- python3 tools/pmc_target.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tise_toolbox_amd import device  # noqa: E402

dev = torch.device("cuda", 0)
B = 500
# sizes large enough that inputs do not sit in the 256 MiB Infinity Cache between launches:
# rotate over 6 image batches (590 MB) and 6 feature batches
imgs = [torch.randint(0, 256, (B, 256, 256, 3), dtype=torch.uint8, device=dev) for _ in range(6)]
feats = [torch.rand((B, 2048), device=dev) for _ in range(6)]
lut = device.make_lut(True)
acc = device.StatsAccumulator(2048, dev)
for i in range(12):
    x = device.resize_bilinear_u8(imgs[i % 6], (299, 299), lut, channels_last=True)
    acc.update_parts(feats[i % 6], cov=True, col_sum=True)
    del x
torch.cuda.synchronize()
mu, sigma = acc.finalize()
torch.cuda.synchronize()
print("pmc target done", float(mu.sum()))
