"""CPU restatement of Pillow's 8-bit bilinear resize (test infrastructure).

The reference resizes with ``transforms.Resize((299, 299))`` + ``ToTensor()``
(``image_realism/FID/fid_score.py:208-213``), which is third-party arithmetic:
torchvision 0.9.1 -> Pillow 8.3.2 ``Image.resize(size, BILINEAR)``.  The
algorithm restated here is Pillow's published ``ImagingResample`` for 8-bit
images (src/libImaging/Resample.c): per axis, coefficient windows with
half-pixel centres, normalised in double, rounded to 22-bit fixed point;
horizontal pass -> uint8 -> vertical pass -> uint8.

PINNED bit-exactly against the Pillow installed in the build container
(tests/golden/pil_resize_*.npz, produced by tests/golden/make_golden.py).
"""
import functools

import numpy as np

PRECISION_BITS = 32 - 8 - 2   # Resample.c: 22-bit coefficients


def bilinear_filter(x):
    x = np.abs(x)
    return np.where(x < 1.0, 1.0 - x, 0.0)


def bicubic_filter(x):
    """Resample.c bicubic_filter, a = -0.5 (support 2): clip._transform's Resize(224, BICUBIC)
    (text_relevance/RP_coco.py:31,64; positional_alignment/PA.py:30,34 through clip.load's preprocess)."""
    a = -0.5
    x = np.abs(x)
    return np.where(x < 1.0, ((a + 2.0) * x - (a + 3.0)) * x * x + 1, np.where(x < 2.0, (((x - 5) * x + 8) * x - 4) * a, 0.0))


FILTERS = {"bilinear": (bilinear_filter, 1.0), "bicubic": (bicubic_filter, 2.0)}


@functools.lru_cache(maxsize=256)
def precompute_coeffs(in_size, out_size, filter="bilinear"):
    """Resample.c precompute_coeffs() + normalize_coeffs_8bpc() for the whole axis.

    Returns (bounds int32 [out,2] = (xmin, count), kk int32 [out, ksize]) -- read-only for callers: the tables of a
    (in_size, out_size, filter) triple are computed once per process (the tests resize thousands of equal-sized images).
    """
    bilinear_filter, support = FILTERS[filter]      # (the name below is the filter function of the chosen kind)
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = support * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        x = np.arange(xmax)
        w = bilinear_filter((x + xmin - center + 0.5) * ss)
        ww = w.sum()
        if ww != 0.0:
            w = w / ww
        # normalize_coeffs_8bpc: round half away from zero in double, truncate to int
        q = np.where(w < 0, -0.5 + w * (1 << PRECISION_BITS), 0.5 + w * (1 << PRECISION_BITS))
        kk[xx, :xmax] = q.astype(np.int64).astype(np.int32)
        bounds[xx] = (xmin, xmax)
    bounds.setflags(write=False)
    kk.setflags(write=False)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def _resample_axis0(img, out_size, filter="bilinear"):
    """Resample along axis 0 of an (L, ...) uint8 array."""
    bounds, kk = precompute_coeffs(img.shape[0], out_size, filter)
    # Resample.c ImagingResampleHorizontal_8bpc / Vertical_8bpc: ss = 1 << (PRECISION_BITS - 1); ss += pixel * k over the window;
    # clip8(ss).  Integer arithmetic: the sum does not depend on its order, so all output rows are formed at once -- the window
    # of output xx is gathered at xmin .. xmin + ksize - 1 (indices clamped into the axis; the coefficients behind a window's
    # count are zero, so what is read there does not matter).
    ksize = kk.shape[1]
    idx = np.minimum(bounds[:, :1].astype(np.int64) + np.arange(ksize, dtype=np.int64)[None, :], img.shape[0] - 1)      # (out, ksize)
    src = img.astype(np.int64)
    acc = np.full((out_size,) + img.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
    for x in range(ksize):
        acc += src[idx[:, x]] * kk[:, x].astype(np.int64).reshape((out_size,) + (1,) * (img.ndim - 1))
    return _clip8(acc)


def resize_u8(img, out_h, out_w, filter="bilinear"):
    """(H, W, C) uint8 -> (out_h, out_w, C) uint8, bit-exact to Image.resize((out_w, out_h), BILINEAR | BICUBIC).

    Pillow runs the horizontal pass first (skipped when the width is unchanged),
    stores uint8, then the vertical pass (skipped when the height is unchanged).
    """
    img = np.ascontiguousarray(img)
    h, w = img.shape[:2]
    if w != out_w:
        img = np.swapaxes(_resample_axis0(np.swapaxes(img, 0, 1), out_w, filter), 0, 1)
    if h != out_h:
        img = _resample_axis0(img, out_h, filter)
    return np.ascontiguousarray(img)


def resize_bilinear_u8(img, out_h, out_w):
    return resize_u8(img, out_h, out_w, "bilinear")


def to_tensor(img_u8):
    """transforms.ToTensor(): HWC uint8 -> CHW float32 / 255 (fid_score.py:211)."""
    return np.ascontiguousarray(np.transpose(img_u8, (2, 0, 1))).astype(np.float32) / np.float32(255.0)


# image_realism/FID/inception.py:120-124 -- applied to [0,1] pixels
NORM_SCALE = (0.229 / 0.5, 0.224 / 0.5, 0.225 / 0.5)
NORM_BIAS = ((0.485 - 0.5) / 0.5, (0.456 - 0.5) / 0.5, (0.406 - 0.5) / 0.5)


def normalize_input(x_chw):
    """inception.py:120-124 on an fp32 (3,H,W) or (B,3,H,W) array."""
    x = np.array(x_chw, dtype=np.float32, copy=True)
    for c in range(3):
        x[..., c, :, :] = x[..., c, :, :] * np.float32(NORM_SCALE[c]) + np.float32(NORM_BIAS[c])
    return x
