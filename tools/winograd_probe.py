#!/usr/bin/env python3
"""Decision probe: Winograd F(2x2, 3x3) on the split operands for 35x35x96->96 and 8x8x448->384.

Gate: keep the idea only if the layer gets >= 1.3x faster at <= 2e-6 of the output scale against an fp64 convolution.

What is measured, with the product's own kernels (no new kernel is needed to decide):
  direct     the layer as the trunk runs it today (SplitConv: row-window kernel, 3x3, pad 1), time per launch
  winograd   its MATRIX-CORE part only: 16 GEMMs [tiles x Cin] x [Cin x Cout], one per Winograd position, run as 16 launches
             of the default 1x1 kernel on a split tensor of the transformed tiles, raw fp32 out.  The input transform (B^T d B in
             fp32 per 4x4 tile and channel, then a re-split: ~8 vector instructions per transformed value), the output
             transform (A^T m A, bias, ReLU, re-split) and their traffic are NOT in the time: this is a LOWER BOUND of a fused
             Winograd kernel's time -- if even the bound is not 1.3x under the direct launch, the gate fails by construction.
  error      direct (real kernel) and Winograd (transforms in fp32 torch, the 16 GEMMs by the REAL split kernels, so the
             3-MFMA arithmetic is the product's) against an fp64 convolution, as a fraction of the output scale.
Usage: python tools/winograd_probe.py [batch]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from tise_toolbox_amd import conv_split as cs

dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float64)
G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float64)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float64)


def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def probe(h, w, cin, cout, n):
    g = torch.Generator(device="cpu").manual_seed(h * 1000 + cin)
    x = torch.relu(torch.randn((n, h, w, cin), generator=g)) * (0.5 + torch.rand((1, 1, 1, cin), generator=g))     # ReLU-like, per-channel scales
    wt = torch.randn((cout, cin, 3, 3), generator=g) * (2.0 / (9 * cin)) ** 0.5 * (0.3 + 2 * torch.rand((cout, 1, 1, 1), generator=g))
    bias = torch.zeros(cout)
    xs = cs.split(x.to(dev))                                      # what the layer really reads: F(x) = hi + lo 2^-11
    xm = cs.merge(xs)
    nchk = min(n, 16)
    ref = F.conv2d(xm[:nchk].double().permute(0, 3, 1, 2), wt.double().to(dev), padding=1).permute(0, 2, 3, 1)
    scale = float(ref.abs().max())
    # ---- direct --------------------------------------------------------------------------------------------------
    conv = cs.SplitConv(wt, bias, (1, 1), (1, 1), dev)
    out_raw = torch.empty((n, h, w, cout), dtype=torch.float32, device=dev)
    t_direct = timed(lambda: conv(xs, [(0, cout, out_raw, 0, 1)]))
    out_split = cs.new_split(n, h, w, cout, dev)
    t_direct_split = timed(lambda: conv(xs, [(0, cout, out_split, 0, 0)]))
    # raw mode applies bias + ReLU? mode 1 = raw fp32 of the affine result; compare on the pre-activation where possible
    conv(xs, [(0, cout, out_raw, 0, 1)])
    got = out_raw[:nchk].double()
    err_direct = float((got - ref).abs().max()) / scale
    err_direct_relu = float((torch.relu(got) - torch.relu(ref)).abs().max()) / scale
    # ---- Winograd: transforms in fp32 torch, the 16 GEMMs by the real 1x1 split kernels -----------------------------------
    th, tw = (h + 1) // 2, (w + 1) // 2
    xp = F.pad(xm.permute(0, 3, 1, 2), (1, 1 + (2 * tw - w), 1, 1 + (2 * th - h)))                 # (n, c, 2 th + 2, 2 tw + 2)
    d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                                          # (n, c, th, tw, 4, 4)
    bt = BT.float().to(dev)
    V = torch.einsum("ij,nctujk,lk->nctuil", bt, d, bt)                                             # B^T d B in fp32
    U = torch.einsum("ij,ocjk,lk->iloc", G, wt.double(), G)                                        # (4, 4, cout, cin) fp64
    gemms, Vs, Ms = [], [], []
    for i in range(4):
        for j in range(4):
            gemms.append(cs.SplitConv(U[i, j].float().reshape(cout, cin, 1, 1), bias, (1, 1), (0, 0), dev))
            Vs.append(cs.split(V[..., i, j].permute(0, 2, 3, 1).contiguous()))                     # (n, th, tw, 2 cin)
            Ms.append(torch.empty((n, th, tw, cout), dtype=torch.float32, device=dev))

    def run_gemms():
        for k in range(16):
            gemms[k](Vs[k], [(0, cout, Ms[k], 0, 1)])
    t_gemm = timed(run_gemms)
    # the same matrix work as ONE launch (timing only: the 16 positions stacked along M with one position's weights -- what a
    # batched kernel with grid.z = position would cost; 16 separate launches of ~1 workgroup round each pay a tail per launch)
    v_all = torch.cat(Vs, 1).contiguous()                                                           # (n, 16 th, tw, 2 cin)
    m_all = torch.empty((n, 16 * th, tw, cout), dtype=torch.float32, device=dev)
    t_batched = timed(lambda: gemms[0](v_all, [(0, cout, m_all, 0, 1)]))
    del v_all, m_all
    run_gemms()
    M = torch.stack(Ms, -1).reshape(n, th, tw, cout, 4, 4)[:nchk]
    at = AT.float().to(dev)
    Y = torch.einsum("ij,ntucjk,lk->ntuicl", at, M, at)                                             # (n, th, tw, 2, cout, 2) -> spatial
    Y = Y.permute(0, 1, 3, 2, 5, 4).reshape(nchk, 2 * th, 2 * tw, cout)[:, :h, :w]
    err_wino = float((Y.double() - ref).abs().max()) / scale
    # the same with the transforms in fp64 (what the GEMM arithmetic alone costs)
    V64 = torch.einsum("ij,nctujk,lk->nctuil", BT.to(dev), d[:nchk].double(), BT.to(dev))
    flop = 2.0 * n * h * w * 9 * cin * cout
    flop_w = 2.0 * n * th * tw * 16 * cin * cout
    print(f"{h}x{w}x{cin}->{cout} 3x3, batch {n}: direct {t_direct_split:.3f} ms (raw-out {t_direct:.3f}) = {3 * flop / t_direct_split / 1e9:.0f} TF16; "
          f"Winograd matrix part alone (16 launches, {th * tw} tiles/image, K = {cin}) {t_gemm:.3f} ms = {3 * flop_w / t_gemm / 1e9:.0f} TF16 "
          f"-> upper bound of the speed-up {t_direct_split / t_gemm:.2f}x; as ONE batched launch {t_batched:.3f} ms = {3 * flop_w / t_batched / 1e9:.0f} TF16 "
          f"-> {t_direct_split / t_batched:.2f}x (gate 1.3x; input / output transforms and their traffic NOT counted in either); "
          f"error vs fp64 of the output scale: direct {err_direct:.2e} (after ReLU {err_direct_relu:.2e}), Winograd {err_wino:.2e} (gate 2e-6)", flush=True)


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
    probe(35, 35, 96, 96, n)
    probe(8, 8, 448, 384, n)
    probe(71, 71, 80, 192, max(100, n // 3))       # Conv2d_4a-like (unpadded in the trunk; padded here): the one layer with Cout large enough to amortise transforms
