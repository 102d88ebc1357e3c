/* libtise_png.so -- host-side PNG decoder of the image feed (tise_toolbox_amd/_png_worker.py), plain C, no HIP.
 *
 * Replaces, for the files the toolbox is fed with, ``Image.open(f).convert("RGB")`` of the reference's
 * ``Dataset.__getitem__`` (image_realism/FID/img_data.py:19-25; third-party Pillow 8.3.2 -> libpng-style decode):
 * 8-bit RGB or RGBA, non-interlaced PNGs (what ``Image.fromarray(...).save("x.png")`` and the generators' ``save_image``
 * write).  Anything else -- palette, gray, 16-bit, interlaced, tRNS, JPEG -- returns TISE_PNG_UNSUPPORTED and the worker
 * decodes that file with Pillow itself; a file with ANY doubt about it -- a chunk whose CRC-32 does not match (IHDR, IDAT,
 * IEND and every ancillary chunk walked), a zlib stream of the wrong length or checksum, a filter byte above 4 -- returns
 * TISE_PNG_CORRUPT and Pillow decides (it raises its own error, or decodes what it tolerates: it does not check IDAT CRCs).
 * So a return of TISE_PNG_OK means: every checksum of the file is right and the bytes are Pillow's
 * (tests/test_png_fuzz.py: sanitizer build, seeded mutations, rc 0 => Pillow decodes to the same bytes).  PNG is lossless and its decode is fully
 * specified (RFC 2083: zlib inflate + the five row filters), so the bytes equal Pillow's; RGBA -> RGB drops the alpha
 * byte, which is what Pillow's convert("RGB") does (no blending).  tests/test_host_logic.py compares against Pillow.
 *
 * Why: the GPU boxes give a container 16 CPUs of CFS quota (profiles/r05b_host_decode_probe.txt) and Pillow needs
 * 1.25 ms per 256 x 256 image there (12.8 k images/s with every CPU decoding); half of that is zlib's inflate, the other
 * half Pillow's byte-serial unfilter and its copies.  Here: libdeflate's inflate when the library is present (dlopen,
 * no header needed: three stable entry points), zlib's ``uncompress`` otherwise; unfilter a pixel at a time; one pass
 * from the inflated rows into the caller's ring slot.
 */
#include <dlfcn.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#define TISE_PNG_OK 0
#define TISE_PNG_UNSUPPORTED 1   /* a valid-looking file outside the subset: decode it with Pillow */
#define TISE_PNG_CORRUPT 2       /* let Pillow raise its own error */
#define TISE_PNG_SIZE 3          /* decoded fine, but not the expected height x width (w, h are reported) */
#define TISE_PNG_SCRATCH 4       /* scratch too small */

static inline uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

/* ---- libdeflate through dlopen ------------------------------------------------------------------------------------ */
typedef void* (*ld_alloc_fn)(void);
typedef int (*ld_zlib_fn)(void*, const void*, size_t, void*, size_t, size_t*);
typedef uint32_t (*ld_crc_fn)(uint32_t, const void*, size_t);
static ld_zlib_fn g_ld_zlib = 0;
static ld_crc_fn g_ld_crc = 0;
static void* g_ld_dec = 0;
static int g_ld_state = 0;       /* 0 untried, 1 usable, -1 absent */

static void ld_init(void) {
    if (g_ld_state) return;
    g_ld_state = -1;
    if (getenv("TISE_PNG_ZLIB")) return;                      /* A/B switch: force zlib's inflate */
    void* h = dlopen("libdeflate.so.0", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libdeflate.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    ld_alloc_fn alloc = (ld_alloc_fn)dlsym(h, "libdeflate_alloc_decompressor");
    g_ld_zlib = (ld_zlib_fn)dlsym(h, "libdeflate_zlib_decompress");
    if (!alloc || !g_ld_zlib) return;
    g_ld_crc = (ld_crc_fn)dlsym(h, "libdeflate_crc32");       /* carry-less-multiply CRC-32, ~10 x zlib's; optional */
    g_ld_dec = alloc();
    if (g_ld_dec) g_ld_state = 1;
}

int tise_png_inflate_backend(void) { ld_init(); return g_ld_state == 1 ? 1 : 0; }   /* 1 libdeflate, 0 zlib */

/* CRC-32 of a chunk's type + body (RFC 2083 section 3.4), against the four bytes behind the body. */
static int chunk_crc_ok(const uint8_t* type_and_body, size_t n, const uint8_t* stored) {
    ld_init();
    const uint32_t got = g_ld_crc ? g_ld_crc(0, type_and_body, n) : (uint32_t)crc32(0L, type_and_body, (uInt)n);
    return got == be32(stored);
}

/* ---- unfilter: one row, BPP bytes per pixel, in place; `up` = the previous unfiltered row (NULL for the first) --------- */
static inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    int pa = p - a, pb = p - b, pc = p - c;
    pa = pa < 0 ? -pa : pa; pb = pb < 0 ? -pb : pb; pc = pc < 0 ? -pc : pc;
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

/* Paeth rows a pixel at a time on 16-bit SSE lanes (the dependency chain runs from pixel to pixel: ~10 instructions of
 * latency per PIXEL instead of per byte; the scalar form above took 0.9 of a 256 x 256 image's 1.26 ms).  Formulation:
 * pa = |b - c|, pb = |a - c|, pc = |(b - c) + (a - c)|; nearest = pa <= min(pb, pc) ? a : (pb <= pc ? b : c).
 * The 4-byte loads of a 3-byte pixel read one byte past it: inside the scratch buffer (the next row's filter byte, or
 * the padding behind the last row). */
#if defined(__SSSE3__)
#include <tmmintrin.h>
#define TISE_PNG_SIMD 1
static inline __m128i ld4(const uint8_t* p) { int v; memcpy(&v, p, 4); return _mm_unpacklo_epi8(_mm_cvtsi32_si128(v), _mm_setzero_si128()); }
static inline __m128i paeth_px(__m128i a, __m128i b, __m128i c, __m128i d) {
    __m128i pa = _mm_sub_epi16(b, c), pb = _mm_sub_epi16(a, c);
    __m128i pc = _mm_add_epi16(pa, pb);
    pa = _mm_abs_epi16(pa); pb = _mm_abs_epi16(pb); pc = _mm_abs_epi16(pc);
    const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
    const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb);
    /* a where pa is smallest, else b where pb is, else c */
    __m128i nearest = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
    nearest = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, nearest));
    return _mm_and_si128(_mm_add_epi16(d, nearest), _mm_set1_epi16(0xff));
}
static void paeth_row_simd(uint8_t* cur, const uint8_t* up, size_t n, int bpp) {
    __m128i a = _mm_setzero_si128(), c = _mm_setzero_si128();
    for (size_t i = 0; i + bpp <= n; i += bpp) {
        const __m128i b = ld4(up + i);
        const __m128i d = paeth_px(a, b, c, ld4(cur + i));
        const int v = _mm_cvtsi128_si32(_mm_packus_epi16(d, d));
        memcpy(cur + i, &v, bpp);
        a = d; c = b;
    }
}
#endif

#define DEFINE_UNFILTER(BPP)                                                                                 \
    static int unfilter_row_##BPP(int ft, uint8_t* restrict cur, const uint8_t* restrict up, size_t n) {      \
        size_t i;                                                                                             \
        switch (ft) {                                                                                         \
            case 0: return 0;                                                                                 \
            case 1:                                                                                           \
                for (i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);                          \
                return 0;                                                                                     \
            case 2:                                                                                           \
                if (up) for (i = 0; i < n; ++i) cur[i] = (uint8_t)(cur[i] + up[i]);                           \
                return 0;                                                                                     \
            case 3:                                                                                           \
                if (up) {                                                                                     \
                    for (i = 0; i < BPP; ++i) cur[i] = (uint8_t)(cur[i] + (up[i] >> 1));                      \
                    for (; i < n; ++i) cur[i] = (uint8_t)(cur[i] + ((cur[i - BPP] + up[i]) >> 1));            \
                } else {                                                                                      \
                    for (i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + (cur[i - BPP] >> 1));               \
                }                                                                                             \
                return 0;                                                                                     \
            case 4:                                                                                           \
                if (up && TISE_PNG_USE_SIMD) { paeth_row_simd_call(cur, up, n, BPP); }                       \
                else if (up) {                                                                                \
                    /* a pixel at a time: the BPP channels are independent chains */                          \
                    int a[BPP], c[BPP], k;                                                                    \
                    for (k = 0; k < BPP; ++k) { a[k] = cur[k] = (uint8_t)(cur[k] + up[k]); c[k] = up[k]; }    \
                    for (i = BPP; i + BPP <= n; i += BPP)                                                     \
                        for (k = 0; k < BPP; ++k) {                                                           \
                            const int b = up[i + k];                                                          \
                            const int v = (uint8_t)(cur[i + k] + paeth(a[k], b, c[k]));                       \
                            cur[i + k] = (uint8_t)v;                                                          \
                            a[k] = v; c[k] = b;                                                               \
                        }                                                                                     \
                } else {                                                                                      \
                    for (i = BPP; i < n; ++i) cur[i] = (uint8_t)(cur[i] + cur[i - BPP]);                      \
                }                                                                                             \
                return 0;                                                                                     \
            default: return 1;                                                                                \
        }                                                                                                     \
    }
#ifdef TISE_PNG_SIMD
#define TISE_PNG_USE_SIMD 1
#define paeth_row_simd_call(cur, up, n, bpp) paeth_row_simd(cur, up, n, bpp)
#else
#define TISE_PNG_USE_SIMD 0
#define paeth_row_simd_call(cur, up, n, bpp) ((void)0)
#endif
DEFINE_UNFILTER(3)
DEFINE_UNFILTER(4)

/* Parse the chunk list.  On success: *w, *h, *bpp (3 or 4), and the IDAT payload either in place (*idat, *idat_len: a
 * single IDAT chunk) or gathered into scratch (several).  */
static int png_parse(const uint8_t* f, size_t len, int* w, int* h, int* bpp, const uint8_t** idat, size_t* idat_len,
                     uint8_t* scratch, size_t scratch_bytes, size_t* scratch_used) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    if (len < 8 + 25 || memcmp(f, sig, 8) != 0) return TISE_PNG_UNSUPPORTED;   /* not a PNG (a .jpg): Pillow's business */
    size_t p = 8;
    int have_ihdr = 0, n_idat = 0, seen_iend = 0;
    const uint8_t* first = 0;
    size_t first_len = 0, gathered = 0;
    *scratch_used = 0;
    while (p + 12 <= len) {
        const uint32_t cl = be32(f + p);
        const uint8_t* typ = f + p + 4;
        if ((size_t)cl > len - p - 12) return TISE_PNG_CORRUPT;
        const uint8_t* body = f + p + 8;
        if (!chunk_crc_ok(typ, (size_t)cl + 4, body + cl)) return TISE_PNG_CORRUPT;   /* Pillow raises for header chunks, ignores IDAT's */
        if (!have_ihdr) {
            if (memcmp(typ, "IHDR", 4) != 0 || cl != 13) return TISE_PNG_CORRUPT;
            const uint32_t ww = be32(body), hh = be32(body + 4);
            if (ww == 0 || hh == 0 || ww > 65535 || hh > 65535) return TISE_PNG_UNSUPPORTED;
            if (body[8] != 8 || (body[9] != 2 && body[9] != 6) || body[10] != 0 || body[11] != 0 || body[12] != 0)
                return TISE_PNG_UNSUPPORTED;                       /* bit depth 8, RGB / RGBA, no interlace */
            *w = (int)ww; *h = (int)hh; *bpp = body[9] == 2 ? 3 : 4;
            have_ihdr = 1;
        } else if (memcmp(typ, "IDAT", 4) == 0) {
            if (n_idat == 0) { first = body; first_len = cl; }
            else {
                if (n_idat == 1) {                                 /* second IDAT: start gathering */
                    if (first_len > scratch_bytes) return TISE_PNG_SCRATCH;
                    memcpy(scratch, first, first_len);
                    gathered = first_len;
                }
                if (cl > scratch_bytes - gathered) return TISE_PNG_SCRATCH;
                memcpy(scratch + gathered, body, cl);
                gathered += cl;
            }
            ++n_idat;
        } else if (memcmp(typ, "IEND", 4) == 0) {
            seen_iend = 1;
            break;
        } else if (memcmp(typ, "tRNS", 4) == 0 || memcmp(typ, "PLTE", 4) == 0) {
            return TISE_PNG_UNSUPPORTED;                           /* transparency key / palette: Pillow decides what they mean */
        } else if (!(typ[0] & 0x20)) {
            return TISE_PNG_UNSUPPORTED;                           /* unknown critical chunk */
        }
        p += 12 + (size_t)cl;
    }
    if (!have_ihdr || n_idat == 0 || !seen_iend) return TISE_PNG_CORRUPT;      /* a file cut before IEND: Pillow decides */
    if (n_idat == 1) { *idat = first; *idat_len = first_len; }
    else { *idat = scratch; *idat_len = gathered; *scratch_used = (gathered + 63) & ~(size_t)63; }
    return TISE_PNG_OK;
}

/* Size and pixel format of a PNG file image: 0 and (w, h, channels) for the subset decoded here. */
int tise_png_probe(const uint8_t* file, size_t len, int* w, int* h, int* channels) {
    const uint8_t* idat; size_t il, used;
    uint8_t dummy[8];
    int bpp = 0, rc;
    if (!file || !w || !h || !channels) return TISE_PNG_CORRUPT;
    rc = png_parse(file, len, w, h, &bpp, &idat, &il, dummy, 0, &used);
    if (rc == TISE_PNG_SCRATCH) rc = TISE_PNG_OK;                  /* several IDATs: fine for a probe */
    *channels = bpp;
    return rc;
}

/* Scratch a caller must provide for an h x w image of a file of `file_len` bytes. */
size_t tise_png_scratch_bytes(int h, int w, size_t file_len) {
    return (size_t)h * ((size_t)w * 4 + 1) + file_len + 256;
}

/* Decode `file` (the whole PNG file in memory) into dst[h][w][3] uint8.  Returns TISE_PNG_OK, or a code that tells the
 * caller to hand the file to Pillow (UNSUPPORTED / CORRUPT), or TISE_PNG_SIZE with the file's size in *got_w, *got_h. */
int tise_png_decode_rgb8(const uint8_t* file, size_t len, uint8_t* dst, int h, int w, uint8_t* scratch, size_t scratch_bytes,
                         int* got_w, int* got_h) {
    const uint8_t* idat; size_t idat_len, used;
    int fw = 0, fh = 0, bpp = 0;
    if (!file || !dst || !scratch) return TISE_PNG_CORRUPT;
    int rc = png_parse(file, len, &fw, &fh, &bpp, &idat, &idat_len, scratch, scratch_bytes, &used);
    if (got_w) *got_w = fw;
    if (got_h) *got_h = fh;
    if (rc != TISE_PNG_OK) return rc;
    if (fw != w || fh != h) return TISE_PNG_SIZE;
    const size_t stride = (size_t)w * bpp, raw_len = (size_t)h * (stride + 1);
    if (used + raw_len > scratch_bytes) return TISE_PNG_SCRATCH;
    uint8_t* raw = scratch + used;
    ld_init();
    if (g_ld_state == 1) {
        size_t got = 0;
        if (g_ld_zlib(g_ld_dec, idat, idat_len, raw, raw_len, &got) != 0 || got != raw_len) return TISE_PNG_CORRUPT;
    } else {
        uLongf got = (uLongf)raw_len;
        if (uncompress(raw, &got, idat, (uLong)idat_len) != Z_OK || got != raw_len) return TISE_PNG_CORRUPT;
    }
    const uint8_t* up = 0;
    for (int y = 0; y < h; ++y) {
        uint8_t* row = raw + (size_t)y * (stride + 1);
        const int ft = row[0];
        uint8_t* cur = row + 1;
        const int bad = bpp == 3 ? unfilter_row_3(ft, cur, up, stride) : unfilter_row_4(ft, cur, up, stride);
        if (bad) return TISE_PNG_CORRUPT;
        uint8_t* out = dst + (size_t)y * w * 3;
        if (bpp == 3) memcpy(out, cur, stride);
        else
            for (int x = 0; x < w; ++x) { out[3 * x] = cur[4 * x]; out[3 * x + 1] = cur[4 * x + 1]; out[3 * x + 2] = cur[4 * x + 2]; }
        up = cur;
    }
    return TISE_PNG_OK;
}

/* ---- the device-unfilter feed: inflate only ------------------------------------------------------------------------------
 * A ring slot of the image feed (tise_toolbox_amd/png_ring.py, TISE_PNG_UNFILTER=device) is
 *     [ 64-byte header | payload ]       header byte 0 = mode:  0  payload is h x w x 3 RGB pixels (decoded on the host)
 *                                                               3  payload is h rows of (1 filter byte + w x 3 filtered bytes)
 *                                                               4  the same with 4 bytes per pixel (RGBA)
 * For modes 3 / 4 the worker has only INFLATED the file's zlib stream -- the five row filters of RFC 2083 section 6 (and the
 * RGBA -> RGB drop) run on the GPU (csrc/png_unfilter.hip: tise_png_unfilter_rgb8), so a decode process spends ~0.3 ms per
 * 256 x 256 image instead of ~0.65.  A file whose filtered rows do not fit the slot (an RGBA file in a ring sized for RGB)
 * or whose rows are too long for the kernel's LDS tile is decoded completely here (mode 0). */
#define TISE_PNG_SLOT_HDR 64
#define TISE_PNG_DEVICE_ROW_MAX 8192        /* bytes of one filtered row the device kernel stages (png_unfilter.hip) */

size_t tise_png_slot_bytes(int h, int w, int bpp) {
    /* bpp = 0: pixels only.  8 bytes of slack: the kernel's row staging reads whole dwords. */
    const size_t px = (size_t)h * w * 3, raw = bpp ? (size_t)h * ((size_t)w * bpp + 1) : 0;
    const size_t pay = (px > raw ? px : raw) + 8;
    return TISE_PNG_SLOT_HDR + ((pay + 63) & ~(size_t)63);
}

int tise_png_inflate_slot(const uint8_t* file, size_t len, uint8_t* slot, size_t slot_bytes, int h, int w,
                          uint8_t* scratch, size_t scratch_bytes, int* got_w, int* got_h, int* mode_out) {
    const uint8_t* idat; size_t idat_len, used;
    int fw = 0, fh = 0, bpp = 0;
    if (!file || !slot || !scratch || slot_bytes < TISE_PNG_SLOT_HDR) return TISE_PNG_CORRUPT;
    int rc = png_parse(file, len, &fw, &fh, &bpp, &idat, &idat_len, scratch, scratch_bytes, &used);
    if (got_w) *got_w = fw;
    if (got_h) *got_h = fh;
    if (rc != TISE_PNG_OK) return rc;
    if (fw != w || fh != h) return TISE_PNG_SIZE;
    const size_t stride = (size_t)w * bpp, raw_len = (size_t)h * (stride + 1);
    uint8_t* pay = slot + TISE_PNG_SLOT_HDR;
    memset(slot, 0, TISE_PNG_SLOT_HDR);
    if (TISE_PNG_SLOT_HDR + raw_len + 8 > slot_bytes || stride + 1 > TISE_PNG_DEVICE_ROW_MAX) {
        if (TISE_PNG_SLOT_HDR + (size_t)h * w * 3 > slot_bytes) return TISE_PNG_SCRATCH;
        rc = tise_png_decode_rgb8(file, len, pay, h, w, scratch, scratch_bytes, got_w, got_h);
        if (mode_out) *mode_out = 0;
        return rc;
    }
    ld_init();
    if (g_ld_state == 1) {
        size_t got = 0;
        if (g_ld_zlib(g_ld_dec, idat, idat_len, pay, raw_len, &got) != 0 || got != raw_len) return TISE_PNG_CORRUPT;
    } else {
        uLongf got = (uLongf)raw_len;
        if (uncompress(pay, &got, idat, (uLong)idat_len) != Z_OK || got != raw_len) return TISE_PNG_CORRUPT;
    }
    for (int y = 0; y < h; ++y)
        if (pay[(size_t)y * (stride + 1)] > 4) return TISE_PNG_CORRUPT;     /* the kernel trusts the filter bytes */
    slot[0] = (uint8_t)bpp;
    if (mode_out) *mode_out = bpp;
    return TISE_PNG_OK;
}
