"""CPU: the host half of row a2 (csrc/png_decode.c) and its checker (oracle/png_oracle.py).

* the oracle's row-filter restatement == Pillow on files with every filter type (pins the oracle);
* tise_png_inflate_slot writes [header | filtered rows] whose reconstruction by the oracle == Pillow (what the device kernel
  is held to in tests/test_gpu_png.py); RGBA files in RGB-sized slots are decoded completely (mode 0);
* chunk CRC-32s are verified (IHDR / IDAT / IEND / ancillary): a flipped bit anywhere -> TISE_PNG_CORRUPT, the caller hands
  the file to Pillow (three IHDR-CRC-corrupt files used to be accepted);
* a sanitizer build (gcc -fsanitize=address,undefined) of the one C file runs >= 5000 seeded mutations on exact-size heap
  buffers: no report, and native rc 0 => Pillow decodes the same file to the same bytes."""
import ctypes
import io
import os
import shutil
import struct
import subprocess
import zlib

import numpy as np
import pytest

from tests import _png_cases

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PNG_OK, PNG_UNSUPPORTED, PNG_CORRUPT, PNG_SIZE, PNG_SCRATCH = range(5)


def _pillow(blob, mode="RGB"):
    from PIL import Image
    return np.asarray(Image.open(io.BytesIO(blob)).convert(mode))


@pytest.fixture(scope="module")
def lib():
    from tise_toolbox_amd import _png_worker, build
    build.build_png(verbose=False)
    lib = _png_worker.load_decoder()
    assert lib is not None
    return lib


def _inflate(lib, blob, h, w, bpp_ring):
    sb = int(lib.tise_png_slot_bytes(h, w, bpp_ring))
    slot = np.full(sb, 0xee, dtype=np.uint8)
    scratch = np.zeros(int(lib.tise_png_scratch_bytes(h, w, len(blob))), dtype=np.uint8)
    gw, gh, mode = ctypes.c_int(), ctypes.c_int(), ctypes.c_int(-1)
    rc = lib.tise_png_inflate_slot(blob, len(blob), slot.ctypes.data, sb, h, w, scratch.ctypes.data, scratch.size,
                                   ctypes.byref(gw), ctypes.byref(gh), ctypes.byref(mode))
    return rc, mode.value, slot


def _decode(lib, blob, h, w):
    dst = np.zeros((h, w, 3), dtype=np.uint8)
    scratch = np.zeros(int(lib.tise_png_scratch_bytes(h, w, len(blob))), dtype=np.uint8)
    gw, gh = ctypes.c_int(), ctypes.c_int()
    rc = lib.tise_png_decode_rgb8(blob, len(blob), dst.ctypes.data, h, w, scratch.ctypes.data, scratch.size, ctypes.byref(gw), ctypes.byref(gh))
    return rc, dst


@pytest.mark.parametrize("h,w,bpp", [(1, 1, 3), (1, 1, 4), (3, 5, 3), (17, 9, 4), (40, 31, 3)])
def test_oracle_unfilter_equals_pillow_and_slots_reconstruct(lib, h, w, bpp):
    from oracle import png_oracle
    rng = np.random.default_rng(h * 31 + w + bpp)
    img = rng.integers(0, 256, (h, w, bpp), dtype=np.uint8)
    for filters in ([0] * h, [1] * h, [2] * h, [3] * h, [4] * h, None, list(rng.integers(0, 5, h))):
        raw = _png_cases.filter_rows(img, filters if filters is not None else [y % 5 for y in range(h)])
        assert np.array_equal(png_oracle.unfilter_rows(raw, h, w, bpp), img)            # the writer and the oracle are inverse
        blob = _png_cases.write_png(img, filters, idat_sizes=[3, 40])
        want = _pillow(blob)
        assert np.array_equal(want, img[:, :, :3])                                        # Pillow agrees with both
        rc, mode, slot = _inflate(lib, blob, h, w, bpp)
        assert (rc, mode) == (PNG_OK, bpp)
        assert np.array_equal(slot[64:64 + raw.size], raw.reshape(-1))                    # the payload IS the filtered rows
        assert np.array_equal(png_oracle.pixels_of_slot(slot, h, w), want)
        rc, px = _decode(lib, blob, h, w)
        assert rc == PNG_OK and np.array_equal(px, want)


def test_slot_modes_and_sizes(lib):
    from oracle import png_oracle
    h, w = 12, 10
    rng = np.random.default_rng(3)
    rgb, rgba = rng.integers(0, 256, (h, w, 3), dtype=np.uint8), rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    assert int(lib.tise_png_slot_bytes(h, w, 0)) == 64 + 384          # pixels only: 360 + 8 bytes of slack, rounded to 64
    assert int(lib.tise_png_slot_bytes(h, w, 3)) == 64 + 384 and int(lib.tise_png_slot_bytes(h, w, 4)) == 64 + 512
    rc, mode, slot = _inflate(lib, _png_cases.write_png(rgba), h, w, 3)                   # RGBA file, ring sized for RGB
    assert (rc, mode, int(slot[0])) == (PNG_OK, 0, 0)
    assert np.array_equal(png_oracle.pixels_of_slot(slot, h, w), rgba[:, :, :3])
    rc, mode, slot = _inflate(lib, _png_cases.write_png(rgb), h, w, 4)                    # RGB file, ring sized for RGBA
    assert (rc, mode) == (PNG_OK, 3) and np.array_equal(png_oracle.pixels_of_slot(slot, h, w), rgb)
    rc, _, _ = _inflate(lib, _png_cases.write_png(rgb), h, w + 1, 3)
    assert rc == PNG_SIZE
    # a filter-type byte above 4: the device kernel trusts the bytes, so the host refuses the file (Pillow raises on it too)
    raw = _png_cases.filter_rows(rgb, [0] * h)
    raw[5, 0] = 7
    z = zlib.compress(raw.tobytes())
    blob = (b"\x89PNG\r\n\x1a\n" + _png_cases._chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + _png_cases._chunk(b"IDAT", z)
            + _png_cases._chunk(b"IEND", b""))
    assert _inflate(lib, blob, h, w, 3)[0] == PNG_CORRUPT and _decode(lib, blob, h, w)[0] == PNG_CORRUPT
    # rows beyond the kernel's LDS tile (8192 bytes): decoded completely on the host
    wide = rng.integers(0, 256, (2, 2800, 3), dtype=np.uint8)
    rc, mode, slot = _inflate(lib, _png_cases.write_png(wide), 2, 2800, 3)
    assert (rc, mode) == (PNG_OK, 0) and np.array_equal(png_oracle.pixels_of_slot(slot, 2, 2800), wide)


def test_chunk_crcs_are_verified(lib):
    h, w = 9, 11
    img = np.random.default_rng(8).integers(0, 256, (h, w, 3), dtype=np.uint8)
    blob = _png_cases.write_png(img, None, idat_sizes=[20], extra_chunks=[(b"tEXt", b"key\0value")])
    assert _decode(lib, blob, h, w)[0] == PNG_OK
    # walk the chunks; flip one bit in each chunk's body (or, for the empty IEND, in its CRC) and in each stored CRC
    pos, spots = 8, []
    while pos < len(blob):
        n = struct.unpack(">I", blob[pos:pos + 4])[0]
        spots.append((blob[pos + 4:pos + 8], pos + 8 if n else pos + 8 + n, pos + 8 + n))
        pos += 12 + n
    assert [s[0] for s in spots] == [b"IHDR", b"tEXt", b"IDAT", b"IDAT", b"IEND"]
    from PIL import Image
    for typ, body_at, crc_at in spots:
        for at in {body_at, crc_at + 3}:
            bad = bytearray(blob)
            bad[at] ^= 0x10
            bad = bytes(bad)
            if typ == b"IHDR" and at == body_at:
                continue                                                   # changes the declared width: a different (valid-CRC-less) file; covered by the fuzz test
            assert _decode(lib, bad, h, w)[0] == PNG_CORRUPT, (typ, at)
            assert _inflate(lib, bad, h, w, 3)[0] == PNG_CORRUPT, (typ, at)
    bad = bytearray(blob)
    bad[spots[0][2]] ^= 1                                                  # IHDR CRC: Pillow refuses the file -- and so do we now
    with pytest.raises(Exception):
        Image.open(io.BytesIO(bytes(bad))).load()
    assert _decode(lib, bytes(bad), h, w)[0] == PNG_CORRUPT
    assert _decode(lib, blob[:-12], h, w)[0] == PNG_CORRUPT               # cut before IEND


@pytest.mark.timeout(900)
def test_sanitizer_fuzz_rc0_implies_pillow_bytes(tmp_path):
    """the ASan + UBSan build of csrc/png_decode.c on seeded mutations (byte flips, truncations, insertions,
    chunk-length edits) of RGB / RGBA seeds with all filter types, several IDATs, ancillary chunks.  The harness
    (tests/png_fuzz_harness.c) uses exact-size heap buffers.  Assertions: no sanitizer report (exit code 0, empty stderr);
    for every case either entry point returning 0 means Pillow decodes that very file to the same pixels, and a slot that
    holds filtered rows reconstructs (oracle) to them too."""
    from PIL import Image
    from oracle import png_oracle
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    exe = tmp_path / "png_fuzz"
    cmd = [gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-mssse3", "-msse4.1",
           os.path.join(ROOT, "tests", "png_fuzz_harness.c"), os.path.join(ROOT, "tise_toolbox_amd", "csrc", "png_decode.c"),
           "-o", str(exe), "-lz", "-ldl"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0 and ("asan" in r.stderr.lower() or "sanitize" in r.stderr.lower()):
        pytest.skip(f"sanitizer runtime not available: {r.stderr[:200]}")
    assert r.returncode == 0, r.stderr
    h, w = 24, 20
    rng = np.random.default_rng(2024)
    smooth = (np.add.outer(np.arange(h) * 5, np.arange(w) * 3)[..., None] + np.arange(4) * 50).astype(np.uint8)
    noise = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
    buf = io.BytesIO()
    Image.fromarray(smooth[:, :, :3]).save(buf, "PNG")
    seeds = [_png_cases.write_png(smooth[:, :, :3]), _png_cases.write_png(noise[:, :, :3], [4] * h, idat_sizes=[30, 30, 30]),
             _png_cases.write_png(smooth, None, extra_chunks=[(b"tEXt", b"a\0b")]), _png_cases.write_png(noise, list(rng.integers(0, 5, h))),
             buf.getvalue(), _png_cases.write_png(noise[:, :, :3], [3] * h, level=0)]
    n_total, n_ok, n_ok_filtered = 0, 0, 0
    for bpp_ring in (3, 4):
        cases = list(seeds)
        while len(cases) < 2600:
            s = bytearray(seeds[int(rng.integers(len(seeds)))])
            kind = int(rng.integers(6))
            if kind == 0:                                           # flip 1-3 bits anywhere
                for _ in range(int(rng.integers(1, 4))):
                    s[int(rng.integers(len(s)))] ^= 1 << int(rng.integers(8))
            elif kind == 1:                                         # overwrite a byte
                s[int(rng.integers(len(s)))] = int(rng.integers(256))
            elif kind == 2:                                         # truncate
                s = s[:int(rng.integers(len(s)))]
            elif kind == 3:                                         # insert random bytes
                at = int(rng.integers(len(s)))
                s[at:at] = bytes(rng.integers(0, 256, int(rng.integers(1, 9)), dtype=np.uint8))
            elif kind == 4:                                         # edit a chunk length field and repair nothing
                at = 8
                for _ in range(int(rng.integers(0, 4))):
                    n = struct.unpack(">I", bytes(s[at:at + 4]))[0] if at + 4 <= len(s) else 0
                    if at + 12 + n >= len(s):
                        break
                    at += 12 + n
                if at + 4 <= len(s):
                    s[at:at + 4] = struct.pack(">I", int(rng.integers(0, 1 << int(rng.integers(1, 32)))))
            else:                                                   # corrupt the zlib stream but keep the chunk CRC valid
                at = s.find(b"IDAT")
                if at > 0:
                    n = struct.unpack(">I", bytes(s[at - 4:at]))[0]
                    if n > 2 and at + 4 + n + 4 <= len(s):
                        s[at + 4 + int(rng.integers(n))] ^= 1 << int(rng.integers(8))
                        s[at + 4 + n:at + 8 + n] = struct.pack(">I", zlib.crc32(bytes(s[at:at + 4 + n])) & 0xffffffff)
            cases.append(bytes(s))
        inp, outp = tmp_path / f"in{bpp_ring}.bin", tmp_path / f"out{bpp_ring}.bin"
        with open(inp, "wb") as f:
            for c in cases:
                f.write(struct.pack("<I", len(c)) + c)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
        r = subprocess.run([str(exe), str(inp), str(outp), str(h), str(w), str(bpp_ring)], capture_output=True, text=True, env=env, timeout=600)
        assert r.returncode == 0 and not r.stderr.strip(), r.stderr[-3000:]
        data = open(outp, "rb").read()
        sb = 64 + (((max(h * w * 3, h * (w * bpp_ring + 1)) + 8) + 63) & ~63)
        rec = 12 + h * w * 3 + sb
        assert len(data) == rec * len(cases)
        for i, c in enumerate(cases):
            rc1, rc2, mode = struct.unpack("<iii", data[i * rec:i * rec + 12])
            n_total += 1
            if rc1 != 0 and rc2 != 0:
                continue
            want = np.asarray(Image.open(io.BytesIO(c)).convert("RGB"))          # must not raise: rc 0 => Pillow accepts the file
            assert want.shape == (h, w, 3)
            if rc1 == 0:
                n_ok += 1
                assert np.array_equal(np.frombuffer(data, np.uint8, h * w * 3, i * rec + 12).reshape(h, w, 3), want), i
            if rc2 == 0:
                slot = np.frombuffer(data, np.uint8, sb, i * rec + 12 + h * w * 3)
                assert int(slot[0]) == mode
                n_ok_filtered += mode != 0
                assert np.array_equal(png_oracle.pixels_of_slot(slot, h, w), want), i
            assert (rc1 == 0) == (rc2 == 0), (i, rc1, rc2)
    print("fuzz:", n_total, n_ok, n_ok_filtered)
    assert n_total >= 5000 and n_ok >= 12 and n_ok_filtered >= 10, (n_total, n_ok, n_ok_filtered)
