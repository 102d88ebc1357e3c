/* Sanitizer harness of csrc/png_decode.c (tests/test_png_host.py builds it with gcc -fsanitize=address,undefined together
 * with that file; tests only).  Input file: cases of [u32 length][file image]; for every case it calls the three entry
 * points on EXACT-size heap buffers (so that AddressSanitizer sees any byte read or written outside them) and writes
 * [i32 rc_decode][i32 rc_slot][i32 mode][h*w*3 decoded pixels][slot] to the output file. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int tise_png_probe(const uint8_t*, size_t, int*, int*, int*);
size_t tise_png_scratch_bytes(int, int, size_t);
int tise_png_decode_rgb8(const uint8_t*, size_t, uint8_t*, int, int, uint8_t*, size_t, int*, int*);
size_t tise_png_slot_bytes(int, int, int);
int tise_png_inflate_slot(const uint8_t*, size_t, uint8_t*, size_t, int, int, uint8_t*, size_t, int*, int*, int*);

int main(int argc, char** argv) {
    if (argc < 6) return 2;
    FILE* in = fopen(argv[1], "rb");
    FILE* out = fopen(argv[2], "wb");
    const int h = atoi(argv[3]), w = atoi(argv[4]), bpp = atoi(argv[5]);
    if (!in || !out) return 2;
    const size_t sb = tise_png_slot_bytes(h, w, bpp);
    uint32_t len;
    while (fread(&len, 4, 1, in) == 1) {
        uint8_t* file = (uint8_t*)malloc(len ? len : 1);
        if (len && fread(file, 1, len, in) != len) return 3;
        int pw = 0, ph = 0, pc = 0;
        (void)tise_png_probe(file, len, &pw, &ph, &pc);
        const size_t scr = tise_png_scratch_bytes(h, w, len);
        uint8_t* scratch = (uint8_t*)malloc(scr);
        uint8_t* dst = (uint8_t*)calloc((size_t)h * w * 3, 1);
        uint8_t* slot = (uint8_t*)calloc(sb, 1);
        int gw = 0, gh = 0, mode = -1;
        const int32_t rc1 = tise_png_decode_rgb8(file, len, dst, h, w, scratch, scr, &gw, &gh);
        const int32_t rc2 = tise_png_inflate_slot(file, len, slot, sb, h, w, scratch, scr, &gw, &gh, &mode);
        const int32_t m = mode;
        fwrite(&rc1, 4, 1, out); fwrite(&rc2, 4, 1, out); fwrite(&m, 4, 1, out);
        fwrite(dst, 1, (size_t)h * w * 3, out);
        fwrite(slot, 1, sb, out);
        free(file); free(scratch); free(dst); free(slot);
    }
    fclose(in); fclose(out);
    return 0;
}
