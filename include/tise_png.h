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
#define TISE_PNG_CORRUPT 2       /* malformed: let Pillow raise its own error */
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

#ifdef __cplusplus
}
#endif
#endif
