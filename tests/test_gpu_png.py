"""Row a2 on the device: the PNG row filters reversed in HBM (csrc/png_unfilter.hip, tise_png_unfilter_rgb8) behind the
host's inflate (csrc/png_decode.c, tise_png_inflate_slot) == Pillow's ``Image.open(f).convert("RGB")``
(image_realism/FID/img_data.py:19-25), byte for byte: all five filter types, RGB and RGBA, split IDATs, 1 x 1 and odd
widths, heights that are not a multiple of the kernel's 64-row block, host-decoded slots mixed in; then the whole feed
(png_ring.PngRingLoader with the device unfilter) against Pillow and against the host-unfilter feed."""
import ctypes
import io
import os

import numpy as np
import pytest
import torch

from tests import _png_cases

pytestmark = pytest.mark.gpu


def _pillow_rgb(blob):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(blob)).convert("RGB"))


def _inflate_slots(blobs, h, w, bpp_ring):
    """Host half of the feed on a list of file images: (n, slot_bytes) uint8 slots + the modes the library chose."""
    from tise_toolbox_amd import _png_worker
    lib = _png_worker.load_decoder()
    assert lib is not None, "libtise_png.so missing"
    sb = int(lib.tise_png_slot_bytes(h, w, bpp_ring))
    slots = np.zeros((len(blobs), sb), dtype=np.uint8)
    scratch = np.zeros(int(lib.tise_png_scratch_bytes(h, w, max(len(b) for b in blobs))) + 1024, dtype=np.uint8)
    gw, gh, mode = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    modes = []
    for i, blob in enumerate(blobs):
        rc = lib.tise_png_inflate_slot(blob, len(blob), slots[i].ctypes.data, sb, h, w, scratch.ctypes.data, scratch.size,
                                       ctypes.byref(gw), ctypes.byref(gh), ctypes.byref(mode))
        assert rc == 0, (i, rc)
        modes.append(mode.value)
    return slots, modes


def _device_unfilter(slots, h, w):
    from tise_toolbox_amd import _lib
    dev = torch.device("cuda", 0)
    s = torch.from_numpy(slots).to(dev)
    out = torch.full((slots.shape[0], h, w, 3), 0x5a, dtype=torch.uint8, device=dev)
    _lib.call("tise_png_unfilter_rgb8", ctypes.c_void_p(s.data_ptr()), slots.shape[0], slots.shape[1], h, w,
              ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    return out.cpu().numpy()


SIZES = [(1, 1), (1, 7), (7, 1), (5, 3), (64, 64), (65, 33), (130, 67), (256, 256), (200, 301), (63, 1024),
         (12, 2047)]          # the widest rows the device path takes: 4 * 2047 + 1 = 8189 of TISE_PNG_DEVICE_ROW_MAX = 8192 bytes


@pytest.mark.parametrize("h,w", SIZES)
@pytest.mark.parametrize("bpp", [3, 4])
def test_device_unfilter_equals_pillow(h, w, bpp):
    rng = np.random.default_rng(h * 1000 + w + bpp)
    blobs = []
    base = rng.integers(0, 256, (h, w, bpp), dtype=np.uint8)
    smooth = (np.add.outer(np.arange(h) * 3, np.arange(w) * 2)[..., None] + np.arange(bpp) * 40).astype(np.uint8)
    for ft in range(5):                                            # every filter type on every row
        blobs.append(_png_cases.write_png(base, [ft] * h))
    blobs.append(_png_cases.write_png(base, list(rng.integers(0, 5, h))))               # mixed, random per row
    blobs.append(_png_cases.write_png(smooth, None, idat_sizes=[1, 2, 3, 50, 7]))      # y % 5, zlib stream cut into many IDATs
    blobs.append(_png_cases.write_png(smooth, [4] * h, extra_chunks=[(b"tEXt", b"k\0v"), (b"pHYs", bytes(9))]))
    want = np.stack([_pillow_rgb(b) for b in blobs])
    slots, modes = _inflate_slots(blobs, h, w, bpp)
    assert modes == [bpp] * len(blobs)                             # all of them took the device road
    got = _device_unfilter(slots, h, w)
    assert np.array_equal(got, want), np.argwhere((got != want).reshape(len(blobs), -1).any(1)).ravel()


def test_rows_beyond_the_device_limit_take_the_host_decode_and_the_copy_mode():
    """A filtered row longer than TISE_PNG_DEVICE_ROW_MAX (8192 bytes: RGBA at w = 2100 has 8401) is decoded completely on the
    host (csrc/png_decode.c: tise_png_inflate_slot falls back to mode 0, pixels in the slot) and the kernel only copies it; an
    RGB file of that width (6301 bytes) beside it in the same launch still takes the device filters."""
    h, w = 6, 2100
    rng = np.random.default_rng(77)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    blobs = [_png_cases.write_png(rgba, [4, 3, 2, 1, 0, 4]), _png_cases.write_png(rgb, [4, 3, 2, 1, 0, 4])]
    want = np.stack([_pillow_rgb(b) for b in blobs])
    slots, modes = _inflate_slots(blobs, h, w, 4)
    assert modes == [0, 3], modes
    got = _device_unfilter(slots, h, w)
    assert np.array_equal(got, want)


def test_mixed_ring_rgba_in_rgb_slots_and_pillow_files():
    """A ring sized for RGB files: an RGBA file's filtered rows do not fit, the library decodes it completely (mode 0) and the
    kernel copies it; RGB slots beside it are unfiltered on the device; in a ring sized for RGBA both forms fit."""
    from PIL import Image
    h, w = 97, 83
    rng = np.random.default_rng(5)
    rgb = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(rgb).save(buf, "PNG")                          # Pillow's own writer (adaptive filters)
    blobs = [_png_cases.write_png(rgb), _png_cases.write_png(rgba, [4] * h), buf.getvalue(), _png_cases.write_png(rgba)]
    want = np.stack([_pillow_rgb(b) for b in blobs])
    slots, modes = _inflate_slots(blobs, h, w, 3)
    assert modes == [3, 0, 3, 0]
    assert np.array_equal(_device_unfilter(slots, h, w), want)
    slots, modes = _inflate_slots(blobs, h, w, 4)
    assert modes == [3, 4, 3, 4]
    assert np.array_equal(_device_unfilter(slots, h, w), want)


def test_argument_checks():
    from tise_toolbox_amd import _lib
    lib = _lib.load()
    assert lib.tise_png_unfilter_rgb8(None, 1, 1024, 4, 4, None, None) == _lib.TISE_ERR_INVALID_ARG
    assert lib.tise_png_unfilter_rgb8(None, 0, 1024, 4, 4, None, None) == _lib.TISE_OK
    assert lib.tise_png_unfilter_rgb8(ctypes.c_void_p(4096), 1, 30, 4, 4, ctypes.c_void_p(4096), None) == _lib.TISE_ERR_INVALID_ARG


@pytest.mark.timeout(600)
def test_ring_feed_device_unfilter_equals_pillow_and_host_feed(tmp_path, monkeypatch):
    """The whole feed: PNG files -> inflate-only workers -> ring -> H2D -> device unfilter, against Pillow's pixels and
    against the same loader with TISE_PNG_UNFILTER=host; Pillow-written files (adaptive filters), hand-written ones with
    every filter type, an RGBA file and a palette file (decoded by Pillow in the worker) in one directory."""
    from PIL import Image
    from tests import _cases
    from tise_toolbox_amd import png_ring
    h = w = 96
    imgs = _cases.smooth_images(40, h, w, seed=9)
    files = []
    for i in range(40):
        f = tmp_path / f"{i:04d}.png"
        if i % 4 == 0:
            f.write_bytes(_png_cases.write_png(imgs[i], [(i // 4 + y) % 5 for y in range(h)], idat_sizes=[100, 1000]))
        elif i == 13:
            rgba = np.concatenate([imgs[i], np.full((h, w, 1), 77, np.uint8)], axis=2)
            f.write_bytes(_png_cases.write_png(rgba))
        elif i == 22:
            Image.fromarray(imgs[i]).convert("P").save(f)
        else:
            Image.fromarray(imgs[i]).save(f)
        files.append(str(f))
    want = np.stack([np.asarray(Image.open(f).convert("RGB")) for f in files])
    dev = torch.device("cuda", 0)
    for mode in ("device", "host"):
        monkeypatch.setenv("TISE_PNG_UNFILTER", mode)
        for bs, group, workers in ((5, 3, 3), (1, 40, 2), (8, 1, 4)):
            ld = png_ring.PngRingLoader(files, bs, dev, group=group, workers=workers, chunk=4)
            assert ld.framed == (mode == "device")
            got = torch.cat([b.clone() for b in ld]).cpu().numpy()
            n = (40 // bs) * bs
            assert got.shape[0] == n and np.array_equal(got, want[:n]), (mode, bs, group)
