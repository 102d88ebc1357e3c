"""Deterministic synthetic cases shared by the golden generator and the tests.

Everything here is seeded numpy; the d=2048 Frechet cases are stored in
tests/golden only as the reference's scalar and regenerated from these functions.
"""
import numpy as np


def pool3_like_features(n, d, seed, latent=None, shift=0.0, noise=0.02):
    """Non-negative, correlated, pool3-looking features: relu(Z W + b) + small noise, float32."""
    rng = np.random.default_rng(seed)
    latent = latent or max(2, min(64, d // 2))
    z = rng.standard_normal((n, latent))
    w = np.random.default_rng(1000 + d).standard_normal((latent, d)) / np.sqrt(latent)
    b = np.random.default_rng(2000 + d).standard_normal(d) * 0.3 + shift
    x = np.maximum(z @ w + b, 0.0) * 0.4
    x = x + noise * np.abs(rng.standard_normal((n, d)))
    return x.astype(np.float32)


def stats(x):
    x = np.asarray(x, dtype=np.float64)
    return np.mean(x, axis=0), np.cov(x, rowvar=False)


def frechet_case(d, kind, seed=0):
    """(mu1, sigma1, mu2, sigma2) float64 for small-d golden vectors."""
    if kind == "fullrank":
        m1, s1 = stats(pool3_like_features(6 * d, d, seed))
        m2, s2 = stats(pool3_like_features(5 * d, d, seed + 1, shift=0.15))
    elif kind == "rankdef":
        n = max(3, d // 2)                       # N < d: both covariances singular
        m1, s1 = stats(pool3_like_features(n, d, seed))
        m2, s2 = stats(pool3_like_features(n + 1, d, seed + 1, shift=0.15))
    elif kind == "identical":
        m1, s1 = stats(pool3_like_features(6 * d, d, seed))
        m2, s2 = m1.copy(), s1.copy()
    elif kind == "shifted":
        m1, s1 = stats(pool3_like_features(6 * d, d, seed))
        m2, s2 = m1 + 0.25, s1.copy()
    else:
        raise ValueError(kind)
    return m1, s1, m2, s2


def frechet_case_2048(kind, n1, n2):
    x1 = pool3_like_features(n1, 2048, 42)
    x2 = pool3_like_features(n2, 2048, 43, shift=0.1)
    m1, s1 = stats(x1)
    m2, s2 = stats(x2)
    return m1, s1, m2, s2


def smooth_images(n, h=256, w=256, seed=0, shift=0.0):
    """'MS-COCO-shaped' smooth synthetic uint8 images: per image a sum of
    K=8 random-orientation sinusoids / gaussian blobs per channel; seed = base seed + index."""
    out = np.empty((n, h, w, 3), dtype=np.uint8)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    yy /= h
    xx /= w
    for i in range(n):
        rng = np.random.default_rng(seed * 1000003 + i)
        img = np.zeros((h, w, 3), np.float32)
        for c in range(3):
            acc = np.zeros((h, w), np.float32)
            for _ in range(4):
                th = rng.uniform(0, np.pi)
                f = rng.uniform(1.0, 12.0)
                ph = rng.uniform(0, 2 * np.pi)
                acc += rng.uniform(0.2, 1.0) * np.sin(2 * np.pi * f * (np.cos(th) * xx + np.sin(th) * yy) + ph)
            for _ in range(4):
                cx, cy, s = rng.uniform(0, 1), rng.uniform(0, 1), rng.uniform(0.03, 0.3)
                acc += rng.uniform(-1.5, 1.5) * np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * s * s))
            img[..., c] = acc
        img = (img - img.min()) / (img.max() - img.min() + 1e-6)
        img = np.clip(img * (0.8 + shift) + 0.1 * rng.uniform(), 0, 1)
        out[i] = (img * 255.0 + 0.5).astype(np.uint8)
    return out
