"""CPU restatement of the reference FID statistics layer (test infrastructure).

Every function cites the lines of ``/root/reference/image_realism/FID/fid_score.py``
it follows.  PINNED: ``tests/test_oracle_golden.py`` checks each function
against golden vectors produced by the reference's own functions
(``tests/golden/make_golden.py``).
"""
import numpy as np
from scipy import linalg


def n_used_images(n_images, batch_size):
    """Drop-last bookkeeping of ``DataLoader(drop_last=True)`` + ``get_activations``.

    fid_score.py:215-217 (drop_last=True => len(loader) = N // bs) and
    fid_score.py:90-96 (d0 = len(loader) * bs; n_used = (d0 // bs) * bs).
    """
    n_batches = n_images // batch_size
    return n_batches * batch_size


def get_activations(batches, forward, batch_size, dims):
    """fid_score.py:67-118 with the model call abstracted as ``forward``.

    ``batches`` is a sized iterable of batches (the DataLoader), ``forward``
    maps a batch to an array (B, dims) or (B, dims, h, w).  Returns a float64
    array (n_used, dims): fp32 activations widened on assignment (:98,:113),
    spatial maps reduced with a global average (:110-111).
    """
    d0 = len(batches) * batch_size                         # :90
    if batch_size > d0:                                    # :91-93
        batch_size = d0
    n_batches = d0 // batch_size                           # :95 (ZeroDivisionError when d0 == 0, as the reference)
    n_used = n_batches * batch_size                        # :96
    pred_arr = np.empty((n_used, dims))                    # :98  (float64)
    for i, batch in enumerate(batches):                    # :99
        start = i * batch_size
        end = start + batch_size
        pred = np.asarray(forward(batch))
        if pred.ndim == 4 and (pred.shape[2] != 1 or pred.shape[3] != 1):
            pred = pred.mean(axis=(2, 3), dtype=np.float32)  # :110-111 adaptive_avg_pool2d (fp32)
        pred_arr[start:end] = pred.reshape(batch_size, -1)   # :113
    return pred_arr


def calculate_activation_statistics(act):
    """fid_score.py:193-196: mu = mean(act, 0); sigma = cov(act, rowvar=False)."""
    act = np.asarray(act, dtype=np.float64)
    mu = np.mean(act, axis=0)
    sigma = np.cov(act, rowvar=False)
    return mu, sigma


def statistics_from_sums(n, s, S):
    """The additive form the device path accumulates: n, s = sum x, S = sum x x^T.

    Algebraically equal to fid_score.py:194-195 (np.cov uses ddof=1):
    sigma = (S - n mu mu^T) / (n - 1).
    """
    mu = s / n
    sigma = (S - n * np.outer(mu, mu)) / (n - 1)
    return mu, sigma


class ImaginaryComponentError(ValueError):
    pass


def calculate_frechet_distance(mu1, sigma1, mu2, sigma2, eps=1e-6):
    """fid_score.py:121-171, statement by statement (scipy.linalg.sqrtm form)."""
    mu1 = np.atleast_1d(mu1)                               # :143-144
    mu2 = np.atleast_1d(mu2)
    sigma1 = np.atleast_2d(sigma1)                         # :146-147
    sigma2 = np.atleast_2d(sigma2)
    assert mu1.shape == mu2.shape, "Training and test mean vectors have different lengths"          # :149
    assert sigma1.shape == sigma2.shape, "Training and test covariances have different dimensions"  # :150
    diff = mu1 - mu2                                       # :152
    covmean = linalg.sqrtm(sigma1.dot(sigma2))             # :155 (disp=False only suppresses the warning)
    if not np.isfinite(covmean).all():                     # :156-160
        offset = np.eye(sigma1.shape[0]) * eps
        covmean = linalg.sqrtm((sigma1 + offset).dot(sigma2 + offset))
    if np.iscomplexobj(covmean):                           # :163-167
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            m = np.max(np.abs(covmean.imag))
            raise ImaginaryComponentError("Imaginary component {}".format(m))
        covmean = covmean.real
    tr_covmean = np.trace(covmean)                         # :169
    return diff.dot(diff) + np.trace(sigma1) + np.trace(sigma2) - 2 * tr_covmean  # :171


def calculate_frechet_distance_symmetric(mu1, sigma1, mu2, sigma2):
    """Symmetric-eigenvalue form of the same quantity.

    Tr sqrt(S1 S2) = sum_i sqrt(lambda_i(S1^{1/2} S2 S1^{1/2})).  This is the
    formulation the HIP kernels implement; kept here so tests can separate
    "formulation vs reference" error from "kernel vs formulation" error.
    """
    mu1 = np.atleast_1d(mu1).astype(np.float64)
    mu2 = np.atleast_1d(mu2).astype(np.float64)
    s1 = np.atleast_2d(sigma1).astype(np.float64)
    s2 = np.atleast_2d(sigma2).astype(np.float64)
    w, v = np.linalg.eigh((s1 + s1.T) * 0.5)
    w = np.clip(w, 0.0, None)
    s1h = (v * np.sqrt(w)) @ v.T
    m = s1h @ s2 @ s1h
    lam = np.linalg.eigvalsh((m + m.T) * 0.5)
    tr_covmean = np.sqrt(np.clip(lam, 0.0, None)).sum()
    diff = mu1 - mu2
    return diff.dot(diff) + np.trace(s1) + np.trace(s2) - 2 * tr_covmean
