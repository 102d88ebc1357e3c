#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference and Pillow):

    python tests/golden/make_golden.py

What is produced (all small .npz files, inputs + the reference's outputs):

  G1 frechet_*.npz      (mu1, sigma1, mu2, sigma2) -> fid, computed by the reference's
                        own ``calculate_frechet_distance`` (image_realism/FID/fid_score.py:121-171)
                        imported with a stubbed ``torchvision`` (absent here; the module only
                        needs it at import time).
  G2 actstats_*.npz     a fake model + fake loader pushed through the reference's
                        ``calculate_activation_statistics`` / ``get_activations`` (:67-118,:174-196):
                        drop-last bookkeeping, fp32->fp64 widening, adaptive_avg_pool branch.
  G3 is_reduce_*.npz    logits + the scores of the IS* reductions.  The reference modules
                        (inception_score_star_coco.py, object_centric_inception_score.py,
                        inception_score_star_bird.py) cannot be imported (TensorFlow / run at
                        import), so these come from the statement-by-statement restatement in
                        oracle/is_oracle.py -- self-generated, documented as such.
  G4 pil_resize_*.npz   uint8 images and Pillow's ``Image.resize((299,299), BILINEAR)`` output --
                        the third-party arithmetic behind ``transforms.Resize`` (fid_score.py:210).
  G5 frechet_d2048_*.npz  d=2048 cases stored as low-rank FACTORS (regenerated deterministically
                        by tests/_cases.py) with the reference scalar only.

The reference sources are imported, never copied: nothing from /root/reference is written
into the repository except numbers.
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True

REF_FID_DIR = "/root/reference/image_realism/FID"


def import_reference_fid_score():
    """Import the reference fid_score.py with empty stand-in modules for torchvision."""
    class _InceptionStub:                      # argparse only reads this class attribute
        BLOCK_INDEX_BY_DIM = {64: 0, 192: 1, 768: 2, 2048: 3}
    for name in ("torchvision", "torchvision.models", "torchvision.transforms"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.path.insert(0, REF_FID_DIR)
    argv, sys.argv = sys.argv, [sys.argv[0]]
    try:
        import fid_score as ref          # noqa: the reference module
    finally:
        sys.argv = argv
        sys.path.remove(REF_FID_DIR)
    return ref


def main():
    from tests import _cases
    ref = import_reference_fid_score()
    import torch

    # ---- G1: Frechet distance, small d ------------------------------------------------------
    for d in (8, 64, 192):
        for kind in ("fullrank", "rankdef", "identical", "shifted"):
            mu1, s1, mu2, s2 = _cases.frechet_case(d, kind, seed=d)
            fid = ref.calculate_frechet_distance(mu1, s1, mu2, s2)
            np.savez_compressed(os.path.join(HERE, f"frechet_d{d}_{kind}.npz"),
                                mu1=mu1, sigma1=s1, mu2=mu2, sigma2=s2, fid=np.float64(fid))
            print(f"G1 d={d} {kind}: fid={fid!r}")

    # ---- G5: d = 2048, stored as factors ---------------------------------------------------
    for kind, n1, n2 in (("fullrank", 3000, 2600), ("rankdef", 1000, 1000)):
        mu1, s1, mu2, s2 = _cases.frechet_case_2048(kind, n1, n2)
        fid = ref.calculate_frechet_distance(mu1, s1, mu2, s2)
        np.savez_compressed(os.path.join(HERE, f"frechet_d2048_{kind}.npz"),
                            kind=kind, n1=n1, n2=n2, fid=np.float64(fid),
                            trace1=np.trace(s1), trace2=np.trace(s2),
                            sigma1_probe=s1[:4, :4].copy(), sigma2_probe=s2[:4, :4].copy())
        print(f"G5 {kind}: fid={fid!r}")

    # ---- G2: activation statistics through the reference's own loop -------------------------
    class FakeModel(torch.nn.Module):
        """Linear map of the batch -> [B, d, 2, 2] so the adaptive_avg_pool branch runs."""
        def __init__(self, d, seed):
            super().__init__()
            g = torch.Generator().manual_seed(seed)
            self.w = torch.nn.Parameter(torch.randn(12, d * 4, generator=g), requires_grad=False)
            self.d = d
        def forward(self, x):
            return [torch.relu(x.reshape(x.shape[0], -1) @ self.w).reshape(x.shape[0], self.d, 2, 2)]

    d, bs, n = 24, 5, 37
    g = torch.Generator().manual_seed(7)
    data = torch.rand(n, 3, 2, 2, generator=g)
    n_batches = n // bs                       # DataLoader(drop_last=True)
    loader = [data[i * bs:(i + 1) * bs] for i in range(n_batches)]
    model = FakeModel(d, 3)
    act = ref.get_activations(loader, model, bs, d, False, False)
    mu, sigma = ref.calculate_activation_statistics(loader, model, bs, d, False, False)
    feats32 = np.concatenate([torch.nn.functional.adaptive_avg_pool2d(model(b)[0], (1, 1)).numpy().reshape(bs, -1)
                              for b in loader], 0)
    np.savez_compressed(os.path.join(HERE, "actstats_fake_model.npz"),
                        data=data.numpy(), w=model.w.numpy(), batch_size=bs, dims=d,
                        feats32=feats32, act=act, mu=mu, sigma=sigma)
    print("G2 act", act.shape, act.dtype, "n_used", act.shape[0], "of", n)

    # ---- G3: IS* reductions (restated, see docstring) ---------------------------------------
    from oracle import is_oracle
    rng = np.random.default_rng(11)
    for name, n, c, T, rule, drop in (("coco", 203, 1000, is_oracle.T_COCO, "coco", False),
                                      ("ois", 157, 80, is_oracle.T_OIS, "ois", False),
                                      ("bird", 130, 51, is_oracle.T_BIRD, "coco", True)):
        logits = (rng.standard_normal((n, c)) * 3.0).astype(np.float32)
        logits[:, : max(1, c // 50)] += 4.0
        m32, s32 = is_oracle.inception_score_from_logits(logits, T, 10, rule, drop, dtype=np.float32)
        m64, s64 = is_oracle.inception_score_from_logits(logits, T, 10, rule, drop, dtype=np.float64)
        np.savez_compressed(os.path.join(HERE, f"is_reduce_{name}.npz"), logits=logits, temperature=T,
                            splits=10, rule=rule, drop_first=drop, mean32=m32, std32=s32, mean64=m64, std64=s64)
        print(f"G3 {name}: fp32 {m32!r} {s32!r}  fp64 {m64!r} {s64!r}")

    # ---- G4: Pillow bilinear resize ----------------------------------------------------------
    from PIL import Image
    rng = np.random.default_rng(5)
    imgs = {
        "rand256": rng.integers(0, 256, (256, 256, 3), dtype=np.uint8),
        "grad256": np.stack([np.add.outer(np.arange(256), np.arange(256)) // 2,
                             np.tile(np.arange(256), (256, 1)),
                             np.tile(np.arange(256)[:, None], (1, 256))], -1).astype(np.uint8),
        "const256": np.full((256, 256, 3), 200, np.uint8),
        "rand_64x48": rng.integers(0, 256, (64, 48, 3), dtype=np.uint8),
        "rand_500x375": rng.integers(0, 256, (500, 375, 3), dtype=np.uint8),
        "rand_300x299": rng.integers(0, 256, (300, 299, 3), dtype=np.uint8),
    }
    out = {}
    for k, im in imgs.items():
        out["in_" + k] = im
        out["out_" + k] = np.asarray(Image.fromarray(im).resize((299, 299), Image.BILINEAR))
    import PIL
    np.savez_compressed(os.path.join(HERE, "pil_resize_299.npz"), pillow_version=PIL.__version__, **out)
    print("G4", list(imgs))


if __name__ == "__main__":
    main()
