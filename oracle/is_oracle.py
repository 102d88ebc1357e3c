"""CPU restatement of the reference IS* reductions (test infrastructure).

The three reference modules cannot be imported as they are (TensorFlow, or they
run the whole evaluation at import), so each function below follows the cited
lines statement by statement.  PINNED: tests/golden/make_golden_is.py runs the
three reference scripts by path under stub tensorflow / torchvision modules and
stores logits + the scores/text the reference computed (is_ref_*.npz);
tests/test_oracle_golden.py::test_is_oracle_matches_reference_script_run checks
this file against them.  Inputs are LOGITS; the temperature/softmax the
reference applies inside its graph is restated in ``softmax_with_temperature``.
"""
import numpy as np
from scipy.stats import entropy

T_COCO = 0.9091363549232483   # image_realism/IS/coco/inception_score_star_coco.py:107
T_BIRD = 0.5980541706085205   # image_realism/IS/bird/inception_score_star_bird.py:192
T_OIS = 2.1737587451934814    # object_fidelity/O-IS/object_centric_inception_score.py:55


def coco_text(mean, std):
    """inception_score_star_coco.py:153-154 result-file text."""
    return "[Inception Score] mean: {:.5f} std: {:.5f}".format(mean, std)


def bird_text(mean, std):
    """inception_score_star_bird.py:208-209."""
    return f"IS = {mean}  +-  {std}"


def ois_text(mean, std):
    """object_centric_inception_score.py:126-127."""
    return f"O-IS: {mean} +-  {std}"


def softmax_with_temperature(logits, temperature, dtype=np.float32, drop_first_class=False):
    """inception_score_star_coco.py:107-108 (tf.div(logits, T); tf.nn.softmax) in ``dtype``.

    ``drop_first_class`` restates inception_score_star_bird.py:189
    (tf.slice(logits, [0, 1], ...): class 0 is the unused background).
    """
    z = np.asarray(logits, dtype=dtype)
    if drop_first_class:
        z = z[:, 1:]
    z = z / dtype(temperature)
    z = z - z.max(axis=1, keepdims=True)
    e = np.exp(z)
    return (e / e.sum(axis=1, keepdims=True)).astype(dtype)


def split_bounds_coco(n, splits):
    """inception_score_star_coco.py:55 / inception_score_star_bird.py:99-100."""
    return [(i * n // splits, (i + 1) * n // splits) for i in range(splits)]


def split_bounds_ois(n, splits):
    """object_centric_inception_score.py:73 (remainder N mod splits is dropped)."""
    return [(k * (n // splits), (k + 1) * (n // splits)) for k in range(splits)]


def inception_score_coco(preds, splits=10):
    """inception_score_star_coco.py:52-60 (same statements as bird :97-108).

    ``preds`` (N, C) softmax outputs, fp32 in the reference.  Returns
    (mean, std) as Python floats (np.std => ddof 0).
    """
    preds = np.asarray(preds)
    scores = []
    for i in range(splits):
        part = preds[(i * preds.shape[0] // splits):((i + 1) * preds.shape[0] // splits), :]
        kl = part * (np.log(part) - np.log(np.expand_dims(np.mean(part, 0), 0)))
        kl = np.mean(np.sum(kl, 1))
        scores.append(np.exp(kl))
    return np.mean(scores).item(), np.std(scores).item()


def inception_score_ois(preds, splits=10):
    """object_centric_inception_score.py:69-81 (fp64 preds, per-row scipy entropy)."""
    preds = np.asarray(preds, dtype=np.float64)            # :60 np.zeros((N, C)) is float64
    n = preds.shape[0]
    split_scores = []
    for k in range(splits):
        part = preds[k * (n // splits):(k + 1) * (n // splits), :]
        py = np.mean(part, axis=0)
        scores = []
        for i in range(part.shape[0]):
            pyx = part[i, :]
            scores.append(entropy(pyx, py))
        split_scores.append(np.exp(np.mean(scores)))
    return np.mean(split_scores), np.std(split_scores)


def inception_score_from_logits(logits, temperature, splits=10, rule="coco",
                                drop_first_class=False, dtype=np.float64):
    """Logits -> score with the whole reduction evaluated in ``dtype``.

    dtype=float32 is the arithmetic the coco/bird reference runs; dtype=float64
    is the exact-arithmetic reading the device kernels are compared with
    (|dIS| <= 1e-4 budget, north_star).
    """
    preds = softmax_with_temperature(logits, temperature, dtype=dtype,
                                     drop_first_class=drop_first_class)
    if rule == "coco":
        return inception_score_coco(preds, splits)
    if rule == "ois":
        m, s = inception_score_ois(preds, splits)
        return float(m), float(s)
    raise ValueError(rule)


def is_sums(logits, temperature, idx_base, n_total, splits, rule="coco", drop_first_class=False):
    """The additive per-split sums the device kernel accumulates.

    A_k = sum_{i in split k} sum_c p_ic log p_ic ;  B_kc = sum_{i in split k} p_ic
    for the rows with global indices idx_base .. idx_base+rows-1, in float64.
    """
    z = np.asarray(logits, dtype=np.float64)
    if drop_first_class:
        z = z[:, 1:]
    z = z / temperature
    z = z - z.max(axis=1, keepdims=True)
    lse = np.log(np.exp(z).sum(axis=1, keepdims=True))
    logp = z - lse
    p = np.exp(logp)
    bounds = split_bounds_coco(n_total, splits) if rule == "coco" else split_bounds_ois(n_total, splits)
    A = np.zeros(splits)
    B = np.zeros((splits, z.shape[1]))
    for k, (lo, hi) in enumerate(bounds):
        a = max(lo - idx_base, 0)
        b = min(hi - idx_base, z.shape[0])
        if b > a:
            A[k] = (p[a:b] * logp[a:b]).sum()
            B[k] = p[a:b].sum(axis=0)
    return A, B


def is_finalize(A, B, n_total, splits, rule="coco"):
    """score_k = exp(A_k/n_k - sum_c pbar_c log pbar_c); mean/std with ddof 0."""
    bounds = split_bounds_coco(n_total, splits) if rule == "coco" else split_bounds_ois(n_total, splits)
    scores = []
    for k, (lo, hi) in enumerate(bounds):
        nk = hi - lo
        pbar = B[k] / nk
        nz = pbar > 0
        scores.append(np.exp(A[k] / nk - (pbar[nz] * np.log(pbar[nz])).sum()))
    return float(np.mean(scores)), float(np.std(scores))
