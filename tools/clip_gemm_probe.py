#!/usr/bin/env python3
"""csrc/clip_ops.hip gemm_f16_kernel against torch.matmul (hipBLASLt / rocBLAS) on the GEMM shapes of the CLIP ViT-B/32
towers at the batch sizes RP_coco uses (text 2048 x 77 tokens, image 512 x 50 tokens)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tise_toolbox_amd import clip_hip

dev = torch.device("cuda:0")


def timed(fn, it=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


shapes = [("text qkv", 2048 * 77, 1536, 512), ("text out", 2048 * 77, 512, 512), ("text fc", 2048 * 77, 2048, 512), ("text proj", 2048 * 77, 512, 2048),
          ("image qkv", 512 * 50, 2304, 768), ("image out", 512 * 50, 768, 768), ("image fc", 512 * 50, 3072, 768), ("image proj", 512 * 50, 768, 3072),
          ("patch embed", 512 * 49, 768, 3072)]
for name, m, n, k in shapes:
    a = torch.randn((m, k), device=dev, dtype=torch.float16)
    w = torch.randn((n, k), device=dev, dtype=torch.float16) * 0.05
    b = torch.randn(n, device=dev, dtype=torch.float16)
    out = torch.empty((m, n), device=dev, dtype=torch.float16)
    ms_h = timed(lambda: clip_hip.gemm(a, w, b, out=out))
    ref = torch.empty((m, n), device=dev, dtype=torch.float16)
    ms_t = timed(lambda: torch.addmm(b, a, w.t(), out=ref))
    err = (out.float() - ref.float()).abs().max().item() / ref.float().abs().max().item()
    fl = 2.0 * m * n * k
    print(f"{name:12s} M={m:6d} N={n:4d} K={k:4d}: clip_ops {ms_h:6.3f} ms {fl / ms_h / 1e9:6.0f} TFLOP/s | torch.addmm {ms_t:6.3f} ms {fl / ms_t / 1e9:6.0f} TFLOP/s | rel diff {err:.1e}", flush=True)
