"""CPU restatement of PNG row-filter reconstruction (RFC 2083 section 6) -- TEST INFRASTRUCTURE ONLY.

The reference decodes a file with ``Image.open(f).convert("RGB")`` (image_realism/FID/img_data.py:19-25); the arithmetic
is third-party (Pillow 8.3.2 -> its PNG decoder: zlib inflate, then the five row filters).  The product splits that decode
in two -- inflate on the host (csrc/png_decode.c), filters on the GPU (csrc/png_unfilter.hip) -- and this module restates
the second half byte by byte so that the host half can be checked without a GPU.  PINNED against Pillow itself in this
container (tests/test_png_host.py: oracle(inflate(file)) == Pillow's pixels on files with every filter type).
Pure Python / numpy loops: small images only."""
import numpy as np

SLOT_HDR = 64       # csrc/png_decode.c: TISE_PNG_SLOT_HDR


def unfilter_rows(raw, h, w, bpp):
    """raw: h rows of (1 filter-type byte + w*bpp filtered bytes) -> (h, w, bpp) uint8.  RFC 2083 6.2-6.6: Recon(x) =
    Filt(x) + {0, Recon(a), Recon(b), floor((Recon(a) + Recon(b)) / 2), PaethPredictor(a, b, c)} mod 256 with a = the byte
    bpp positions to the left, b = the byte above, c = above-left; bytes outside the image are 0."""
    raw = np.asarray(raw, dtype=np.uint8).reshape(h, w * bpp + 1)
    out = np.zeros((h, w * bpp), dtype=np.int64)
    zero = np.zeros(w * bpp, dtype=np.int64)
    for y in range(h):
        ft = int(raw[y, 0])
        f = raw[y, 1:].astype(np.int64)
        up = out[y - 1] if y else zero
        cur = out[y]
        if ft == 0:
            cur[:] = f
        elif ft == 2:
            cur[:] = (f + up) & 255
        elif ft in (1, 3, 4):
            for i in range(w * bpp):
                a = cur[i - bpp] if i >= bpp else 0
                b = up[i]
                c = up[i - bpp] if i >= bpp else 0
                if ft == 1:
                    pred = a
                elif ft == 3:
                    pred = (a + b) >> 1
                else:
                    p = a + b - c
                    pa, pb, pc = abs(p - a), abs(p - b), abs(p - c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[i] = (f[i] + pred) & 255
        else:
            raise ValueError(f"filter type {ft}")
    return out.astype(np.uint8).reshape(h, w, bpp)


def pixels_of_slot(slot, h, w):
    """A ring slot of the device-unfilter feed ([64-byte header | payload], header byte 0 = 0 pixels / 3 / 4 filtered rows)
    -> (h, w, 3) uint8, what tise_png_unfilter_rgb8 must write for it."""
    slot = np.asarray(slot, dtype=np.uint8).reshape(-1)
    mode = int(slot[0])
    pay = slot[SLOT_HDR:]
    if mode == 0:
        return pay[:h * w * 3].reshape(h, w, 3).copy()
    if mode not in (3, 4):
        raise ValueError(f"slot mode {mode}")
    return unfilter_rows(pay[:h * (w * mode + 1)], h, w, mode)[:, :, :3].copy()     # RGBA -> RGB: alpha dropped, no blending
