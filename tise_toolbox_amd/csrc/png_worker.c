/* tise_png_worker -- native decode process of the image feed (tise_toolbox_amd/png_ring.py), plain C, no Python, no HIP.
 *
 * Same protocol as tise_toolbox_amd/_png_worker.py (which remains the FALLBACK worker for files outside this decoder's
 * subset): it attaches to the two shared-memory files inherited from the parent (pixel ring + control block), claims
 * walk-ordered chunks of files under a POSIX record lock on the control file, waits until the chunk's ring slot has been
 * consumed, decodes every file of the chunk into the slot (csrc/png_decode.c, linked in: the whole decode, or inflate only
 * for the device-unfilter slot format) and sets the chunk's ``done`` byte.
 *
 * Why a native program: a Python worker needs ~0.15 s to import numpy + Pillow before its first file -- on a box with 16
 * CPUs of quota sixteen of them burn 2.4 CPU-seconds at start, a quarter of the whole 12 000-file job of bench.py's png_feed
 * leg (profiles/r06c_png_feed_timeline.txt) and a visible part of a CLI's start-up.  This one decodes its first file ~2 ms after
 * exec.
 *
 * A file this decoder does not take (TISE_PNG_UNSUPPORTED: palette / gray / 16-bit / interlaced / JPEG ...; TISE_PNG_CORRUPT;
 * unreadable) makes the worker hand the WHOLE chunk back: done = 3 and header word NEED_PY = 1; the parent then starts
 * Python fallback workers (``_png_worker.py --fallback``) which redo such chunks with Pillow -- so every pixel that does not
 * come from this decoder comes from Pillow itself, and Pillow's own exception text reaches the user for broken files.
 *
 *   tise_png_worker RING_FD CTL_FD RING_SIZE CTL_SIZE
 */
#define _GNU_SOURCE
#include <errno.h>
#include <fcntl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

/* control-block header words (int64), mirrored in _png_worker.py */
enum { HDR_NEXT, HDR_CONSUMED, HDR_STOP, HDR_NCHUNKS, HDR_ERR, HDR_CHUNK, HDR_NSLOTS, HDR_H, HDR_W, HDR_NFILES, HDR_FILES_OFF,
       HDR_DONE_OFF, HDR_ERRTXT_OFF, HDR_STARTED, HDR_RGBONLY, HDR_IMG_BYTES, HDR_NEED_PY, HDR_WORDS = 24 };
#define ERRTXT_BYTES 1024

#define TISE_PNG_OK 0
#define TISE_PNG_UNSUPPORTED 1
#define TISE_PNG_CORRUPT 2
#define TISE_PNG_SIZE 3
#define TISE_PNG_SCRATCH 4

int tise_png_probe(const uint8_t*, size_t, int*, int*, int*);
size_t tise_png_scratch_bytes(int, int, size_t);
int tise_png_decode_rgb8(const uint8_t*, size_t, uint8_t*, int, int, uint8_t*, size_t, int*, int*);
int tise_png_inflate_slot(const uint8_t*, size_t, uint8_t*, size_t, int, int, uint8_t*, size_t, int*, int*, int*);

static void lock_ctl(int fd, int type) {
    struct flock fl;
    memset(&fl, 0, sizeof fl);
    fl.l_type = (short)type; fl.l_whence = SEEK_SET; fl.l_start = 0; fl.l_len = 8;
    while (fcntl(fd, F_SETLKW, &fl) == -1 && errno == EINTR) {}
}

static void nap_us(long us) {
    struct timespec ts = {0, us * 1000L};
    nanosleep(&ts, 0);
}

int main(int argc, char** argv) {
    if (argc < 5) return 2;
    const int ring_fd = atoi(argv[1]), ctl_fd = atoi(argv[2]);
    const size_t ring_size = (size_t)strtoull(argv[3], 0, 10), ctl_size = (size_t)strtoull(argv[4], 0, 10);
    uint8_t* ctl = (uint8_t*)mmap(0, ctl_size, PROT_READ | PROT_WRITE, MAP_SHARED, ctl_fd, 0);
    uint8_t* ring = (uint8_t*)mmap(0, ring_size, PROT_READ | PROT_WRITE, MAP_SHARED, ring_fd, 0);
    if (ctl == MAP_FAILED || ring == MAP_FAILED) return 3;
    volatile int64_t* hdr = (volatile int64_t*)ctl;
    const int64_t n_chunks = hdr[HDR_NCHUNKS], chunk = hdr[HDR_CHUNK], nslots = hdr[HDR_NSLOTS];
    const int h = (int)hdr[HDR_H], w = (int)hdr[HDR_W];
    const int64_t n_files = hdr[HDR_NFILES];
    volatile uint8_t* done = ctl + hdr[HDR_DONE_OFF];
    const int64_t* offs = (const int64_t*)(ctl + hdr[HDR_FILES_OFF]);
    const char* blob = (const char*)(ctl + hdr[HDR_FILES_OFF] + 8 * (n_files + 1));
    const size_t px_bytes = (size_t)h * w * 3;
    const size_t img_bytes = hdr[HDR_IMG_BYTES] ? (size_t)hdr[HDR_IMG_BYTES] : px_bytes;
    const int framed = img_bytes != px_bytes;
    const int rgb_only = hdr[HDR_RGBONLY] != 0;
    uint8_t *file = 0, *scratch = 0;
    size_t file_cap = 0, scratch_cap = 0;
    char name[4096];

    lock_ctl(ctl_fd, F_WRLCK);
    hdr[HDR_STARTED] += 1;
    lock_ctl(ctl_fd, F_UNLCK);
    const pid_t parent = getppid();
    while (!hdr[HDR_STOP] && getppid() == parent) {
        lock_ctl(ctl_fd, F_WRLCK);
        const int64_t c = hdr[HDR_NEXT];
        if (c < n_chunks) hdr[HDR_NEXT] = c + 1;
        lock_ctl(ctl_fd, F_UNLCK);
        if (c >= n_chunks) break;
        while (c >= hdr[HDR_CONSUMED] + nslots) {           /* the slot still holds a chunk the parent has not copied */
            if (hdr[HDR_STOP] || getppid() != parent) return 0;
            nap_us(300);
        }
        const int64_t lo = c * chunk, hi = (c + 1) * chunk < n_files ? (c + 1) * chunk : n_files;
        int hand_back = 0, failed = 0;
        char err[ERRTXT_BYTES];
        for (int64_t i = lo; i < hi && !hand_back && !failed; ++i) {
            const size_t nl = (size_t)(offs[i + 1] - offs[i]);
            if (nl >= sizeof name) { hand_back = 1; break; }
            memcpy(name, blob + offs[i], nl);
            name[nl] = 0;
            const int fd = open(name, O_RDONLY | O_CLOEXEC);
            if (fd < 0) { hand_back = 1; break; }          /* Python reports the I/O error in its own words */
            struct stat st;
            if (fstat(fd, &st) != 0 || st.st_size <= 0) { close(fd); hand_back = 1; break; }
            const size_t len = (size_t)st.st_size;
            if (len + 16 > file_cap) {
                file_cap = len + (len >> 2) + 4096;
                free(file);
                file = (uint8_t*)malloc(file_cap);
                if (!file) return 4;
            }
            size_t got = 0;
            while (got < len) {
                const ssize_t r = read(fd, file + got, len - got);
                if (r < 0 && errno == EINTR) continue;
                if (r <= 0) break;
                got += (size_t)r;
            }
            close(fd);
            if (got != len) { hand_back = 1; break; }
            int gw = 0, gh = 0, pc = 0, mode = 0;
            if (rgb_only && tise_png_probe(file, len, &gw, &gh, &pc) == TISE_PNG_OK && pc != 3) {
                snprintf(err, sizeof err, "ValueError: RAGGED %.700s: not a plain RGB image (the device preprocess needs 3-channel files)", name);
                failed = 1;
                break;
            }
            const size_t need = tise_png_scratch_bytes(h, w, len);
            if (need > scratch_cap) {
                scratch_cap = need + (need >> 2);
                free(scratch);
                scratch = (uint8_t*)malloc(scratch_cap);
                if (!scratch) return 4;
            }
            uint8_t* slot = ring + ((size_t)(c % nslots) * (size_t)chunk + (size_t)(i - lo)) * img_bytes;
            const int rc = framed ? tise_png_inflate_slot(file, len, slot, img_bytes, h, w, scratch, scratch_cap, &gw, &gh, &mode)
                                  : tise_png_decode_rgb8(file, len, slot, h, w, scratch, scratch_cap, &gw, &gh);
            if (rc == TISE_PNG_OK) continue;
            if (rc == TISE_PNG_SIZE) {
                snprintf(err, sizeof err, "ValueError: RAGGED %.700s: %dx%d where the first image is %dx%d", name, gh, gw, h, w);
                failed = 1;
            } else {
                hand_back = 1;                              /* UNSUPPORTED / CORRUPT / SCRATCH: Pillow decides or raises */
            }
        }
        if (failed) {
            lock_ctl(ctl_fd, F_WRLCK);
            if (hdr[HDR_ERR] == 0) {
                err[ERRTXT_BYTES - 1] = 0;
                memcpy(ctl + hdr[HDR_ERRTXT_OFF], err, strlen(err) + 1);
                hdr[HDR_ERR] = c + 1;
            }
            lock_ctl(ctl_fd, F_UNLCK);
            done[c] = 2;
            return 1;
        }
        if (hand_back) {
            hdr[HDR_NEED_PY] = 1;
            __sync_synchronize();
            done[c] = 3;
        } else {
            __sync_synchronize();                           /* the pixels are in the slot before the byte says so */
            done[c] = 1;
        }
    }
    return 0;
}
