"""Where the networks' parameters come from.

The reference always evaluates with PRETRAINED parameters: torchvision's ``inception_v3(pretrained=True)`` download
(image_realism/FID/inception.py:57 -> ``$TORCH_HOME/hub/checkpoints/inception_v3_google-1a9a5a14.pth``), the 80-class
fine-tune ``weights/inceptionv3_fine_to_with_80_coco_classes.pth`` (object_fidelity/O-IS/object_centric_inception_score.py:45,
O-FID/inception.py:63) and ``clip.load("ViT-B/32")`` (text_relevance/RP_coco.py:33 -> ``~/.cache/clip/ViT-B-32.pt``).
There is no network here, so the CLIs resolve, in order:

  1. ``--weights PATH``                       (torchvision / OpenAI-format state_dict)
  2. the path the reference itself would read (the cache file / relative path above), if it exists
  3. ``--synthetic-weights``: seeded stand-in parameters -- throughput and plumbing runs only.  A warning goes to
     stderr and every result line / file carries ``SYNTHETIC_TAG`` so that it cannot pass for a real score
     (``ranking_score --collect`` refuses tagged files).
  4. otherwise: ``RuntimeError`` -- never a silent stand-in.
"""
import os
import sys

SYNTHETIC_TAG = " [synthetic weights: not a real score]"

_KINDS = {
    "inception": ("InceptionV3 (torchvision inception_v3_google-1a9a5a14.pth)",
                  lambda: [os.path.join(_torch_home(), "hub", "checkpoints", "inception_v3_google-1a9a5a14.pth"),
                           os.path.join(_torch_home(), "checkpoints", "inception_v3_google-1a9a5a14.pth")]),
    "inception80": ("80-class fine-tuned InceptionV3",
                    lambda: [os.path.join("weights", "inceptionv3_fine_to_with_80_coco_classes.pth")]),
    "clip": ("CLIP ViT-B/32", lambda: [os.path.expanduser(os.path.join("~", ".cache", "clip", "ViT-B-32.pt"))]),
}


def _torch_home():
    return os.path.expanduser(os.environ.get("TORCH_HOME", os.path.join(os.environ.get("XDG_CACHE_HOME", "~/.cache"), "torch")))


def warn_synthetic(what):
    print(f"[tise] WARNING: {what} runs with SEEDED STAND-IN parameters (no pretrained file given): scores are "
          f"meaningless except for comparing code paths on identical inputs", file=sys.stderr, flush=True)


def resolve(weights, synthetic, kind):
    """-> (path or None, tag).  ``None`` means seeded stand-in parameters (only with ``synthetic``)."""
    what, defaults = _KINDS[kind]
    if synthetic:
        # an explicit request always wins: a seeded plumbing / throughput run must not pick up a file that happens to
        # sit in ~/.cache or the working directory on one machine and not on another
        if weights:
            raise RuntimeError("--synthetic-weights and --weights are mutually exclusive")
        warn_synthetic(what)
        return None, SYNTHETIC_TAG
    if weights:
        if not os.path.exists(weights):
            raise RuntimeError("Invalid path: %s" % weights)
        return weights, ""
    for p in defaults():
        if os.path.exists(p):
            print(f"[tise] {what}: parameters from {p}", file=sys.stderr, flush=True)
            return p, ""
    raise RuntimeError(
        f"no parameters for {what}: the reference downloads them, this machine cannot.  Pass --weights PATH, put the "
        f"file at {defaults()[0]!r}, or pass --synthetic-weights for a plumbing/throughput run with seeded stand-ins")
