"""A small PNG WRITER for the tests of row a2 (tests only): 8-bit RGB / RGBA files with a chosen filter type per row and a
chosen split of the zlib stream into IDAT chunks -- what Pillow's own writer never produces on demand.  Written against
RFC 2083 (sections 3, 6); the decoded pixels of every file are taken from Pillow itself in the tests, never from here."""
import struct
import zlib

import numpy as np


def _chunk(typ, body):
    return struct.pack(">I", len(body)) + typ + body + struct.pack(">I", zlib.crc32(typ + body) & 0xffffffff)


def _paeth(a, b, c):
    p = a.astype(np.int32) + b - c
    pa, pb, pc = np.abs(p - a), np.abs(p - b), np.abs(p - c)
    return np.where((pa <= pb) & (pa <= pc), a, np.where(pb <= pc, b, c))


def filter_rows(img, filters):
    """img (h, w, bpp) uint8, filters: one type 0..4 per row -> the filtered scanlines (h, 1 + w*bpp) uint8."""
    h, w, bpp = img.shape
    raw = img.reshape(h, w * bpp).astype(np.int32)
    out = np.zeros((h, 1 + w * bpp), dtype=np.uint8)
    zero = np.zeros(w * bpp, dtype=np.int32)
    for y in range(h):
        cur = raw[y]
        up = raw[y - 1] if y else zero
        left = np.concatenate([np.zeros(bpp, np.int32), cur[:-bpp]]) if w * bpp > bpp else np.zeros(w * bpp, np.int32)
        upleft = np.concatenate([np.zeros(bpp, np.int32), up[:-bpp]]) if w * bpp > bpp else np.zeros(w * bpp, np.int32)
        ft = int(filters[y])
        pred = {0: zero, 1: left, 2: up, 3: (left + up) >> 1, 4: _paeth(left, up, upleft)}[ft]
        out[y, 0] = ft
        out[y, 1:] = (cur - pred) & 255
    return out


def write_png(img, filters=None, idat_sizes=None, level=6, extra_chunks=()):
    """PNG file bytes of img (h, w, 3 or 4) uint8.  ``filters``: per-row filter types (default: y % 5);
    ``idat_sizes``: the zlib stream is cut into IDAT chunks of these sizes (the rest in a last one; None: a single chunk);
    ``extra_chunks``: (type, body) ancillary chunks placed before the first IDAT."""
    img = np.ascontiguousarray(img, dtype=np.uint8)
    h, w, bpp = img.shape
    assert bpp in (3, 4)
    if filters is None:
        filters = [y % 5 for y in range(h)]
    z = zlib.compress(filter_rows(img, filters).tobytes(), level)
    parts, pos = [], 0
    for n in (idat_sizes or []):
        if pos >= len(z):
            break
        parts.append(z[pos:pos + n])
        pos += n
    if pos < len(z) or not parts:
        parts.append(z[pos:])
    out = b"\x89PNG\r\n\x1a\n" + _chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2 if bpp == 3 else 6, 0, 0, 0))
    for typ, body in extra_chunks:
        out += _chunk(typ, body)
    for part in parts:
        out += _chunk(b"IDAT", part)
    return out + _chunk(b"IEND", b"")
