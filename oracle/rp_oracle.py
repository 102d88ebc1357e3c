"""CPU restatement of the R-precision (RP-COCO) and positional-alignment (PA) reductions.

TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product path (tise_toolbox_amd/ fails loudly
without the HIP library).

Follows the reference's own code around the CLIP towers:
  text_relevance/RP_coco.py:41-52   ten bins over the shuffled item ids, samples_per_bin = int(N / 10), the last bin
                                    takes the remainder
  text_relevance/RP_coco.py:68-80   candidates = [true caption] + mismatched captions; success iff
                                    np.argmax(softmax(logits_per_image)) == 0 (first maximum wins)
  text_relevance/RP_coco.py:79-88   bin score = successes / len(bin); result = mean and population std of the bins
  positional_alignment/PA.py:33-43  success iff softmax([true, false])[0] > 0.6
  positional_alignment/PA.py:52-67  per-phrase success rate, PA = mean over phrases
CLIP itself (third-party `clip` @ git master, README.md:43; ViT-B/32) is absent from /root/reference and from this
image: for the towers parity is UNPINNED.  The reductions are pinned by tests/golden/rp_stub_*.npz / pa_stub.json:
outputs of the reference scripts themselves, run with a stub `clip` module (tests/golden/make_golden_rp.py).
"""
import numpy as np


def clip_logits(img_emb, txt_emb, logit_scale=100.0, normalize=True):
    """logits_per_image of CLIP for one image against its candidates: scale * cosine (fp64)."""
    a = np.asarray(img_emb, dtype=np.float64)
    t = np.asarray(txt_emb, dtype=np.float64)
    if normalize:
        a = a / np.linalg.norm(a)
        t = t / np.linalg.norm(t, axis=-1, keepdims=True)
    return logit_scale * (t @ a)


def clip_forward_probs(img_emb, txt_emb, logit_scale=100.0, normalize=True, dtype=np.float16):
    """What the reference scripts compare (RP_coco.py:72-78, PA.py:37-42): CLIP.forward's
    `logit_scale * image_features @ text_features.t()` (third-party `clip` @ git master, model.py; Python evaluates it
    as (logit_scale * image_features) @ text_features.t()) followed by `.softmax(dim=-1).cpu().numpy()[0]`, with every
    stored tensor in the model's dtype -- fp16 for the model clip.load serves on a GPU, fp32 on the CPU path -- and
    fp32 arithmetic inside each op (torch's opmath for half tensors; accumulation order of the GEMM unspecified:
    float64 here).  Returns the probabilities as `dtype`: np.argmax of them is the first maximum of ROUNDED values."""
    a = np.asarray(img_emb, dtype=dtype)
    t = np.asarray(txt_emb, dtype=dtype)
    if normalize:                                           # x / x.norm(dim=1, keepdim=True) in the model's dtype
        na = np.sqrt((a.astype(np.float64) ** 2).sum()).astype(np.float32).astype(dtype)             # the norm is a tensor too
        nt = np.sqrt((t.astype(np.float64) ** 2).sum(-1, keepdims=True)).astype(np.float32).astype(dtype)
        a = (a.astype(np.float32) / na.astype(np.float32)).astype(dtype)
        t = (t.astype(np.float32) / nt.astype(np.float32)).astype(dtype)
    a = (np.float32(logit_scale) * a.astype(np.float32)).astype(dtype)
    logits = (t.astype(np.float64) @ a.astype(np.float64)).astype(np.float32).astype(dtype)
    z = logits.astype(np.float64) - float(logits.max())
    e = np.exp(z)
    return (e / e.sum()).astype(np.float32).astype(dtype)


def softmax(x):
    x = np.asarray(x, dtype=np.float64)
    e = np.exp(x - x.max())
    return e / e.sum()


def rp_bins(num_captions, perm, num_bins=10):
    """RP_coco.py:41-52.  perm: the shuffled list(range(num_captions))."""
    samples_per_bin = int(len(perm) / num_bins)
    bins = []
    for i in range(num_bins):
        if i == (num_bins - 1) and num_captions % num_bins != 0:
            bins.append(list(perm[i * samples_per_bin:]))
        else:
            bins.append(list(perm[i * samples_per_bin:(i + 1) * samples_per_bin]))
    return bins


def rp_score(success, perm, num_bins=10):
    """success[i] in {0, 1}: whether item i retrieved its true caption.  Returns (mean, std, bin scores)."""
    success = np.asarray(success)
    bins = rp_bins(len(success), list(perm), num_bins)
    scores = [float(np.sum(success[b])) * 1.0 / len(b) for b in bins]
    return float(np.mean(scores)), float(np.std(scores)), scores


def rp_success_from_logits(logits):
    """(N, 1 + mismatched) logits -> success flags (argmax of the softmax == 0)."""
    return np.array([int(np.argmax(softmax(row)) == 0) for row in np.asarray(logits)])


def rp_text(mean, std):
    return f"R-precision: {mean} +- {std}"                  # RP_coco.py:85,90


def pa_score(logits_by_phrase, threshold=0.6):
    """{phrase: (n, 2) logits [true, false]} -> (PA, {phrase: score})."""
    per = {}
    for phrase, rows in logits_by_phrase.items():
        ok = [1.0 if softmax(r)[0] > threshold else 0.0 for r in rows]
        per[phrase] = sum(ok) / len(ok)
    return float(np.mean([per[p] for p in per])), per
