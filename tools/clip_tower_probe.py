#!/usr/bin/env python3
"""Kernel-level breakdown of the hand-written CLIP towers (run under rocprofv3 --kernel-trace --stats)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd import clip_hip, clip_model
dev = torch.device("cuda:0")
model = clip_model.build_clip().to(dev).half()
towers = clip_hip.HipTowers(model)
t = clip_model.HashTokenizer()([f"a photo of item number {i} on the grass" for i in range(2048)]).to(dev)
x = torch.randn((512, 3, 224, 224), device=dev, dtype=torch.float16)
for _ in range(3):
    towers.encode_text(t); towers.encode_image(x)
torch.cuda.synchronize()
