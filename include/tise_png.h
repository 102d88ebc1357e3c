/*
 * libtise_png.so -- host-side PNG decode of the image feed (plain C, gcc; no HIP: the decode worker processes of
 * tise_toolbox_amd/png_ring.py bind it with ctypes and never load the GPU runtime).
 *
 * Replaces ``Image.open(f).convert("RGB")`` of the reference's Dataset.__getitem__ (image_realism/FID/img_data.py:19-25;
 * third-party Pillow 8.3.2) for 8-bit RGB / RGBA non-interlaced PNGs, byte for byte (tests/test_host_logic.py compares
 * with Pillow on every filter type).  Every other file gets a non-zero code and is decoded by Pillow itself.
 */
#ifndef TISE_PNG_H
#define TISE_PNG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define TISE_PNG_OK 0
#define TISE_PNG_UNSUPPORTED 1   /* outside the subset (palette, gray, 16-bit, interlaced, tRNS, not a PNG): use Pillow */
#define TISE_PNG_CORRUPT 2       /* malformed, or ANY doubt: a chunk CRC-32 that does not match (IHDR, IDAT, IEND, ancillary chunks),
                                    a zlib stream of the wrong length / checksum, a filter byte above 4 -- Pillow decides or raises */
#define TISE_PNG_SIZE 3          /* a decodable file of another height x width (reported through got_w / got_h) */
#define TISE_PNG_SCRATCH 4       /* scratch smaller than tise_png_scratch_bytes() */

/* 1: libdeflate was found (dlopen) and does the inflate; 0: zlib's uncompress. */
int tise_png_inflate_backend(void);
/* width, height, channels (3 or 4) of a PNG file image in memory; TISE_PNG_OK only for the subset decoded here. */
int tise_png_probe(const uint8_t* file, size_t len, int* w, int* h, int* channels);
/* scratch bytes tise_png_decode_rgb8 needs for an h x w image held in a file of file_len bytes */
size_t tise_png_scratch_bytes(int h, int w, size_t file_len);
/* decode the PNG file image `file` into dst[h][w][3] (uint8, RGB; an alpha channel is dropped like convert("RGB")) */
int tise_png_decode_rgb8(const uint8_t* file, size_t len, uint8_t* dst, int h, int w, uint8_t* scratch, size_t scratch_bytes,
                         int* got_w, int* got_h);


/* ---- device-unfilter feed (round 6): the host only INFLATES, the GPU reverses the row filters -----------------------------
 * A ring slot is [64-byte header | payload]; header byte 0 = mode: 0 payload = h*w*3 RGB pixels (decoded here), 3 / 4 payload =
 * h rows of (1 filter-type byte + w*3 / w*4 filtered bytes), to be reconstructed by tise_png_unfilter_rgb8 (libtise_hip.so,
 * include/tise_hip.h).  tise_png_slot_bytes: size of a slot for h x w images whose files have `bpp` bytes per pixel (3 RGB,
 * 4 RGBA; 0: pixels only).  tise_png_inflate_slot: parse + CRC-check + inflate `file` into `slot`; a file whose filtered rows do
 * not fit the slot (RGBA in a ring sized for RGB) or exceed the kernel's row limit (8192 bytes) is decoded completely (mode 0).
 * *mode_out receives the mode written.  Return codes as above. */
#define TISE_PNG_SLOT_HDR 64
size_t tise_png_slot_bytes(int h, int w, int bpp);
int tise_png_inflate_slot(const uint8_t* file, size_t len, uint8_t* slot, size_t slot_bytes, int h, int w,
                          uint8_t* scratch, size_t scratch_bytes, int* got_w, int* got_h, int* mode_out);

#ifdef __cplusplus
}
#endif
#endif
