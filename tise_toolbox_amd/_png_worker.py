"""PNG decode worker of tise_toolbox_amd.png_ring.PngRingLoader -- a stand-alone program, started by file path.

It imports numpy and Pillow only (no torch, no package import: a worker is up in ~0.15 s), attaches to two anonymous
shared-memory files inherited from the parent (memfd: the pixel ring and the control block), and decodes
``Image.open(f).convert("RGB")`` -- the reference's ``Dataset.__getitem__`` (image_realism/FID/img_data.py:19-25) --
straight into ring slots, a chunk of consecutive walk-ordered files at a time.  8-bit RGB / RGBA non-interlaced PNGs are
decoded by libtise_png.so (csrc/png_decode.c: libdeflate / zlib inflate + SSE unfilter, the same bytes as Pillow, 2-3x
faster); every other file -- and every file when the library is absent or TISE_PNG_DECODER=pillow -- by Pillow itself.
Two slot formats (header word IMG_BYTES): h * w * 3 = plain RGB pixels (the whole decode happens here), or
tise_png_slot_bytes(h, w, bpp) = the device-unfilter format -- the worker only INFLATES a file into the slot (a 64-byte
header whose first byte names the payload: 0 RGB pixels, 3 / 4 filtered rows of that many bytes per pixel) and the GPU
reverses the row filters (csrc/png_unfilter.hip); a file outside the subset is still decoded here (payload = pixels).

Since round 6 the feed's first-line workers are the native program csrc/png_worker.c (same protocol, up in ~2 ms instead
of the ~0.15 s this file needs to import numpy and Pillow); a chunk holding a file outside the native decoder's subset is
handed back (done byte 3) and THIS program, started with ``--fallback``, redoes such chunks -- the native library again
for the files it takes, Pillow for the rest.  Without ``--fallback`` it is the complete worker it always was
(TISE_PNG_WORKER=python, or no native program built).

Control block (int64 header, see png_ring.HDR_*):
    next_chunk   next chunk to claim (claimed under a POSIX record lock on the control file)
    consumed     chunks the parent has finished with, in order: chunk c may be written once c < consumed + nslots
    stop         the parent asks everyone to leave
    err          first error (chunk + 1); the text sits in the error area
followed by one ``done`` byte per chunk (1 = pixels are in the slot, 2 = failed), the error text, and the file table
(offsets + utf-8 blob).  Nothing is pickled, nothing passes through a queue.
"""
import fcntl
import mmap
import os
import sys
import time

import numpy as np
from PIL import Image

HDR_NEXT, HDR_CONSUMED, HDR_STOP, HDR_NCHUNKS, HDR_ERR, HDR_CHUNK, HDR_NSLOTS, HDR_H, HDR_W, HDR_NFILES, HDR_FILES_OFF, \
    HDR_DONE_OFF, HDR_ERRTXT_OFF, HDR_STARTED, HDR_RGBONLY, HDR_IMG_BYTES, HDR_NEED_PY = range(17)
HDR_WORDS = 24
DONE_OK, DONE_FAILED, DONE_HANDED_BACK, DONE_REDOING = 1, 2, 3, 4     # chunk states (0: not decoded yet); csrc/png_worker.c writes 1 / 2 / 3
SLOT_HDR = 64          # csrc/png_decode.c: TISE_PNG_SLOT_HDR (device-unfilter slots: [64-byte header | payload])
ERRTXT_BYTES = 1024


def load_decoder():
    """ctypes binding of libtise_png.so (next to this file), or None."""
    import ctypes
    if os.environ.get("TISE_PNG_DECODER", "native") == "pillow":
        return None
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libtise_png.so")
    if not os.path.exists(path):
        return None
    try:
        lib = ctypes.CDLL(path)
        lib.tise_png_decode_rgb8.restype = ctypes.c_int
        lib.tise_png_decode_rgb8.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        lib.tise_png_scratch_bytes.restype = ctypes.c_size_t
        lib.tise_png_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_size_t]
        lib.tise_png_probe.restype = ctypes.c_int
        lib.tise_png_probe.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                       ctypes.POINTER(ctypes.c_int)]
        lib.tise_png_inflate_backend.restype = ctypes.c_int
        lib.tise_png_slot_bytes.restype = ctypes.c_size_t
        lib.tise_png_slot_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
        lib.tise_png_inflate_slot.restype = ctypes.c_int
        lib.tise_png_inflate_slot.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_size_t, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int),
                                              ctypes.POINTER(ctypes.c_int)]
        return lib
    except (OSError, AttributeError):
        return None


PNG_OK, PNG_UNSUPPORTED, PNG_CORRUPT, PNG_SIZE, PNG_SCRATCH = range(5)


def main(argv):
    import ctypes
    ring_fd, ctl_fd, ring_size, ctl_size = (int(a) for a in argv[:4])
    ctl = mmap.mmap(ctl_fd, ctl_size)
    ring = mmap.mmap(ring_fd, ring_size)
    hdr = np.frombuffer(ctl, dtype=np.int64, count=HDR_WORDS)
    n_chunks, chunk, nslots = int(hdr[HDR_NCHUNKS]), int(hdr[HDR_CHUNK]), int(hdr[HDR_NSLOTS])
    h, w, n_files = int(hdr[HDR_H]), int(hdr[HDR_W]), int(hdr[HDR_NFILES])
    done = np.frombuffer(ctl, dtype=np.uint8, count=n_chunks, offset=int(hdr[HDR_DONE_OFF]))
    offs = np.frombuffer(ctl, dtype=np.int64, count=n_files + 1, offset=int(hdr[HDR_FILES_OFF]))
    blob_off = int(hdr[HDR_FILES_OFF]) + 8 * (n_files + 1)
    img_bytes = int(hdr[HDR_IMG_BYTES]) or h * w * 3
    framed = img_bytes != h * w * 3                                # device-unfilter slots: header + payload
    slots = np.frombuffer(ring, dtype=np.uint8, count=nslots * chunk * img_bytes).reshape(nslots, chunk, img_bytes)
    ring_addr = slots.ctypes.data
    lib = load_decoder()
    if framed and lib is None:
        raise SystemExit("device-unfilter slots need libtise_png.so")
    mode = ctypes.c_int()
    scratch = np.empty(0, dtype=np.uint8)
    gw, gh = ctypes.c_int(), ctypes.c_int()
    # rgb_only: the consumer resamples the pixels itself and must see what Image.open(f) holds -- clip's preprocess resizes BEFORE
    # it converts to RGB, and Pillow resamples an RGBA image with premultiplied alpha: only 3-channel RGB files may take this road
    rgb_only = bool(hdr[HDR_RGBONLY])
    pc = ctypes.c_int()

    def fail(c, text):
        fcntl.lockf(ctl_fd, fcntl.LOCK_EX, 8, 0)
        try:
            if hdr[HDR_ERR] == 0:
                raw = text.encode("utf-8", "replace")[:ERRTXT_BYTES - 1]
                o = int(hdr[HDR_ERRTXT_OFF])
                ctl[o:o + len(raw) + 1] = raw + b"\0"
                hdr[HDR_ERR] = c + 1
        finally:
            fcntl.lockf(ctl_fd, fcntl.LOCK_UN, 8, 0)
        done[c] = 2

    fcntl.lockf(ctl_fd, fcntl.LOCK_EX, 8, 0)
    hdr[HDR_STARTED] += 1
    fcntl.lockf(ctl_fd, fcntl.LOCK_UN, 8, 0)
    parent = os.getppid()
    fallback = len(argv) > 4 and argv[4] == "--fallback"
    while not hdr[HDR_STOP] and os.getppid() == parent:
        if fallback:
            # redo chunks the native workers handed back (done == 3); leave when every chunk is decoded or failed
            fcntl.lockf(ctl_fd, fcntl.LOCK_EX, 8, 0)
            todo = np.flatnonzero(done == DONE_HANDED_BACK)
            c = int(todo[0]) if todo.size else -1
            if c >= 0:
                done[c] = DONE_REDOING
            fcntl.lockf(ctl_fd, fcntl.LOCK_UN, 8, 0)
            if c < 0:
                if bool(np.all((done == DONE_OK) | (done == DONE_FAILED))) or hdr[HDR_ERR]:
                    break
                time.sleep(0.001)
                continue
        else:
            fcntl.lockf(ctl_fd, fcntl.LOCK_EX, 8, 0)
            c = int(hdr[HDR_NEXT])
            if c < n_chunks:
                hdr[HDR_NEXT] = c + 1
            fcntl.lockf(ctl_fd, fcntl.LOCK_UN, 8, 0)
            if c >= n_chunks:
                break
            while c >= int(hdr[HDR_CONSUMED]) + nslots:               # the slot still holds a chunk the parent has not copied
                if hdr[HDR_STOP] or os.getppid() != parent:            # asked to leave, or the parent is gone
                    return 0
                time.sleep(0.0005)
        dst = slots[c % nslots]
        lo, hi = c * chunk, min((c + 1) * chunk, n_files)
        try:
            for i in range(lo, hi):
                name = bytes(ctl[blob_off + int(offs[i]):blob_off + int(offs[i + 1])]).decode("utf-8", "surrogateescape")
                if lib is not None:
                    with open(name, "rb") as fh_:
                        blob = fh_.read()
                    if rgb_only and lib.tise_png_probe(blob, len(blob), ctypes.byref(gw), ctypes.byref(gh), ctypes.byref(pc)) == PNG_OK \
                            and pc.value != 3:
                        raise ValueError(f"RAGGED {name}: not a plain RGB image (the device preprocess needs 3-channel files)")
                    need = lib.tise_png_scratch_bytes(h, w, len(blob))
                    if scratch.size < need:
                        scratch = np.empty(need + (need >> 2), dtype=np.uint8)
                    slot_addr = ring_addr + ((c % nslots) * chunk + (i - lo)) * img_bytes
                    if framed:
                        rc = lib.tise_png_inflate_slot(blob, len(blob), slot_addr, img_bytes, h, w, scratch.ctypes.data, scratch.size,
                                                       ctypes.byref(gw), ctypes.byref(gh), ctypes.byref(mode))
                    else:
                        rc = lib.tise_png_decode_rgb8(blob, len(blob), slot_addr, h, w,
                                                      scratch.ctypes.data, scratch.size, ctypes.byref(gw), ctypes.byref(gh))
                    if rc == PNG_OK:
                        continue
                    if rc == PNG_SIZE:
                        raise ValueError(f"RAGGED {name}: {gh.value}x{gw.value} where the first image is {h}x{w}")
                    # UNSUPPORTED (palette, gray, 16-bit, interlaced, a JPEG ...) or CORRUPT: Pillow decides / raises
                img = Image.open(name)
                if rgb_only and img.mode != "RGB":
                    raise ValueError(f"RAGGED {name}: mode {img.mode}, not a plain RGB image (the device preprocess needs 3-channel files)")
                if img.size != (w, h):
                    raise ValueError(f"RAGGED {name}: {img.size[1]}x{img.size[0]} where the first image is {h}x{w}")
                img = img.convert("RGB")                          # img_data.py:21
                if framed:
                    dst[i - lo, :SLOT_HDR] = 0                    # mode 0: the payload is pixels
                    dst[i - lo, SLOT_HDR:SLOT_HDR + h * w * 3] = np.asarray(img).reshape(-1)
                else:
                    dst[i - lo, :] = np.asarray(img).reshape(-1)
            done[c] = 1
        except Exception as e:                                    # noqa: BLE001 -- reported to the parent through the control block
            fail(c, f"{type(e).__name__}: {e}")
            return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
