// Sliding-window forms of the split-precision convolution for the 32-channel 3x3 layers of the stem, and the stem layer
// itself (conv_split.hip explains the arithmetic: v ~= hi + lo * 2^-11, three fp16 MFMAs per product, fp32 accumulate; same
// activation layout, same results):
//
//   conv_regw32_kernel   (configuration 34)  register-resident-weights sliding-window kernel for Cin = 32, 3x3, stride 1
//                        (Conv2d_2a, Conv2d_2b)
//   stem_mfma_u8_kernel                       Conv2d_1a (3 -> 32, 3x3 stride 2) straight from the uint8 pixels
//
// Removed, with their measurements kept in profiles/ and DESIGN.md: round 1's 3-stage persistent kernels (configurations
// 0-15, 40-47; within +-3 % of the default kernel); round 2's LDS-resident-weights window kernel (configuration 33; 1.05 ms
// per 500 images of Conv2d_2a against 0.85 for configuration 34); round 3's two re-phased forms of configuration 34
// (configuration 35: two independent four-wave workgroups per CU 1.55 ms, the halves of one workgroup locked in anti-phase
// by the barrier 1.73 ms, against 1.59-1.65 ms for Conv2d_2b: DESIGN 4b).
#include <stdlib.h>
#include <string.h>
#include <hip/hip_fp16.h>
#include "common.h"
#include "conv_epilogue.h"

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef tise_conv_args ConvArgs;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __attribute__((aligned(64))) unsigned char g_pipe_zero_page[64];

#define CP_BK 32

// a tile index beyond the grid (odd tile count, two tiles per iteration): nothing to compute or store (wave-uniform)
__device__ __forceinline__ bool wrap_guard(long long tile, long long ntiles) { return tile < ntiles; }

// ------------------------------------------------------------------------------------------------
// Register-resident-weights sliding-window kernel (configuration 34, round 3) for the same two layers: Conv2d_2a
// (149^2 x 32 -> 32) and Conv2d_2b (147^2 x 32 -> 64, padded).  With Cin = 32 the whole filter of 32 couts is
// 9 taps x 2 cout halves x (hi, lo) = 36 MFMA B-fragments (16 couts x 32 channels each) = 144 VGPRs: every wave keeps the weights of ITS 32 couts in
// registers for its whole life, so the only operand that moves is the input, once: a sliding ring of grid pixels in LDS
// (a tile adds the 128 grid pixels behind the previous window: 16 KB of LDS-DMA per 128-pixel tile, against 51 KB of
// window + 72 KB of weights per tile for the row-window kernel that served Conv2d_2b, where the weight stream alone
// kept the DMA path as busy as the matrix cores).  No weight tile in LDS also means no hand-off buffer is needed:
// all eight waves compute AND run their own epilogue, two per SIMD, so one wave's epilogue runs under the other's MFMAs.
//   wave w: pixel slice (w & 3) of a 128-pixel tile;  COUT = 64: group (w >> 2) owns couts 32*(w >> 2) .. +31 of the
//   SAME tile (both groups read the same A fragments);  COUT = 32: group (w >> 2) owns tile 2*it + (w >> 2).
// Per iteration (one barrier): DMA of the new rows two iterations ahead (2 or 4 pieces per wave), 9 taps x 12 MFMAs (16x16x32) with
// the next tap's A fragments requested first, `s_waitcnt vmcnt` for the rows of the next iteration, epilogue.
// Same K order and MFMA sequence as the generic kernel => bit-identical results.  Grid-pixel tiling, border handling
// and ring arithmetic: the tile is 128 consecutive pixels of the INPUT grid (n, y, x); grid pixels without an output are
// computed and dropped (valid) or their out-of-image taps masked per lane (PADDED).
// Instruction diet (profiles/r03d_conv_mfma_util.md: the first version ran at VALU:MFMA 13, MFMA util 0.33 -- bound by
// vector issue, not by the matrix cores or the DMA): tap masks only in the PADDED instance (Conv2d_2b runs UNPADDED on a
// zero-bordered copy of its input, which Conv2d_2a writes directly: args->out_hp), ring positions and the (n, y, x) of a
// lane's pixels advanced incrementally (no division in the loop), four fragment addresses per tap from one.
// DBG instance only (tools/conv_ablate.py; never launched by the product path): bits 8..10 of args->nseg switch the DMA,
// the taps and the epilogue off.
//
// POOL = true (round 4; Conv2d_2b -> MaxPool2d(3, 2), inception.py:63-65): the max-pool is taken in THIS kernel's epilogue and
// only the pooled split tensor is written (a quarter of the bytes; the pooled-input kernel that used to read the 147^2 x 64
// result back disappears).  The sliding walk already visits the conv rows of an image in order, so pooling needs no halo:
//   * a workgroup owns a contiguous run of POOL ROWS (n, oy) and walks the conv rows 2 oy .. 2 oy + 2 they need, from the
//     start of a grid row (one conv row per workgroup boundary is computed twice: 256 x 149 pixels of 22 million);
//   * vertical first, per lane, no communication: V[x][cout] (fp32, LDS, one 256-byte slot per image column) holds the
//     running maximum of the open window's rows in column x; the lane that converts conv pixel (y, x) does
//       y even, a window ends here:  HV[x] = max(V[x], f), V[x] = f;     y odd: V[x] = max(V[x], f);     first row: V[x] = f
//     (a column is touched by one lane per conv row, and consecutive rows are >= one iteration = one barrier apart);
//   * horizontal after a barrier: thread (ox, 8-cout chunk) takes max(HV[2 ox], HV[2 ox + 1], HV[2 ox + 2]) for the windows
//     whose LAST column arrived in this iteration, re-splits and stores one 16-byte hi and one 16-byte lo chunk.  HV keeps 130
//     columns (slot = x mod 130): the 128 of a tile and the two seam columns of the tile before.
//   V / HV hold the RAW fp32 results; F(r) = hi + lo * 2^-11 of the re-split result -- the value maxpool3s2_split_kernel
//   reads back from the split tensor -- is monotone, so max_i F(r_i) = F(max_i r_i): the pooled maximum is split, merged and
//   split again, and every pooled (hi, lo) pair is BIT-IDENTICAL to pool-after-conv (tests).
//   Slots are swizzled (16-byte chunk ^ pool_swz(slot)) so that the per-lane accesses of the epilogue (4 pixels x 64 bytes
//   per 16 lanes) and the reads of the horizontal pass (2 columns x 8 chunks per 16 lanes) are both conflict-free.
__device__ __forceinline__ int pool_swz(int slot) { return ((slot & 3) << 2) ^ ((slot >> 1) & 1); }
constexpr int POOL_HV_SLOTS = 130;

template <int COUT, bool PADDED, bool DBG = false, bool POOL = false>
__global__ __launch_bounds__(512, 2) void conv_regw32_kernel(const ConvArgs p, const int R16, const long long ntiles) {
    static_assert(!POOL || (COUT == 64 && !PADDED && !DBG), "pooled output: Conv2d_2b's instance only");
    constexpr int KHC = 3, KWC = 3, ntaps = 9;
    constexpr int TPI = 64 / COUT;                        // tiles per iteration
    constexpr int PF = COUT == 64 ? 2 : 1;                // the ring holds the new rows of PF iterations ahead (LDS: 160 KB)
    constexpr int NPW = 2 * TPI;                          // 8-row DMA pieces per wave per iteration
    constexpr int STEP = 128 * TPI;                       // grid pixels per iteration
    constexpr int STG = conv_epi::Staging<1>::BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ms = wave & 3, grp = wave >> 2;
    const int ring = R16 + STEP * PF;                     // rows (a multiple of 16)
    unsigned char* wbuf = lds;
    unsigned char* epi_area = wbuf + ring * 128;
    unsigned char* stage = epi_area + 2048 + wave * STG;      // (POOL: no staging; V and HV live here)
    unsigned char* vbuf = epi_area + 2048;                    // POOL: V[OW][64] fp32
    unsigned char* hvbuf = vbuf + p.OW * 256;                 // POOL: HV[130][64] fp32

    const long long G = (long long)gridDim.x;
    const long long mgrid = (long long)p.N * p.H * p.W;
    const int minoff = -p.PH * p.W - p.PW;
    const unsigned char* xg = reinterpret_cast<const unsigned char*>(p.x);
    const unsigned char* zp = g_pipe_zero_page;
    // a workgroup walks a contiguous run of iterations (TPI tiles each)
    const long long niter_all = (ntiles + TPI - 1) / TPI;
    const long long per = (niter_all + G - 1) / G;
    const long long i_begin = (long long)blockIdx.x * per, i_end = (i_begin + per < niter_all) ? i_begin + per : niter_all;
    if (!POOL && i_begin >= niter_all) return;
    // POOL: this workgroup's pool rows [u0, u1) of the N * OHP -> the grid rows of conv rows 2 oy0 (image n0) .. 2 oy1 + 2 (n1)
    const int ohp = (p.OH - 3) / 2 + 1, owp = (p.OW - 3) / 2 + 1;
    long long g_start = 0;
    unsigned glen = 0;                                    // grid pixels this workgroup walks (POOL)
    if (POOL) {
        const long long units = (long long)p.N * ohp;
        const long long u0 = units * blockIdx.x / G, u1 = units * (blockIdx.x + 1) / G;
        if (u0 >= u1) return;
        const long long n0 = u0 / ohp, n1 = (u1 - 1) / ohp;
        const int oy0 = (int)(u0 - n0 * ohp), oy1 = (int)(u1 - 1 - n1 * ohp);
        g_start = (n0 * p.H + 2 * oy0) * p.W;
        glen = (unsigned)((n1 * p.H + 2 * oy1 + 3) * p.W - g_start);
    }
    const long long t_begin = i_begin * TPI;
    const long long g_base = (POOL ? g_start : t_begin * 128) + minoff;      // grid pixel of relative row 0
    const int nit = POOL ? (int)((glen + 127u) / 128u) : (int)(i_end - i_begin);

    // ---- this wave's weights -> registers -------------------------------------------------------------------
    half8_t bw[ntaps][2][2];                              // [tap][16-cout half][hi / lo]: 16 couts x the tap's 32 channels
    {
        const _Float16* wgt = reinterpret_cast<const _Float16*>(p.w);
        const int cout = (COUT == 64 ? grp * 32 : 0) + (lane & 15);
        const _Float16* wrow = wgt + (long long)cout * p.Kpad + (lane >> 4) * 8;
#pragma unroll
        for (int t = 0; t < ntaps; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                bw[t][s2][0] = *reinterpret_cast<const half8_t*>(wrow + (long long)s2 * 16 * p.Kpad + t * CP_BK);
                bw[t][s2][1] = *reinterpret_cast<const half8_t*>(wrow + (long long)s2 * 16 * p.Kpad + p.w_plane + t * CP_BK);
            }
        // the loads above must be COMPLETE, and known to the compiler to be complete, before the loop: an empty asm that
        // reads and rewrites every fragment makes it wait here.  (Otherwise the wait for these loads is sunk to their
        // first use inside the loop as `s_waitcnt vmcnt(0)`, which every iteration then also spends waiting for the
        // ring rows it has just requested: measured as 0.28 of 1.70 ms with the DMA switched off.)
#pragma unroll
        for (int t = 0; t < ntaps; ++t)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                asm volatile("" : "+v"(bw[t][s2][0]));
                asm volatile("" : "+v"(bw[t][s2][1]));
            }
    }
// DMA of the 8-row pieces Q0, Q0 + QS, ... < NPIECES of the rows that start at relative row REL0 (a long long), whose
// physical ring row is PROW0 (< ring; REL0 mod ring, kept incrementally by the caller)
#define RW32_ROWS(REL0, PROW0, NPIECES, Q0, QS)                                                           \
    {                                                                                                     \
        for (int q = (Q0); q < (NPIECES); q += (QS)) {                                                     \
            const long long g = g_base + (REL0) + q * 8 + (lane >> 3);                                     \
            const bool ok = g >= 0 && g < mgrid;                                                           \
            int prow = (PROW0) + q * 8;                                                                    \
            while (prow >= ring) prow -= ring;                /* scalar; at most twice */                  \
            const int c = (lane & 7) ^ ((lane >> 4) << 1);    /* tise_lds_swz of a row inside an 8-row piece */ \
            const unsigned char* src = xg + g * 128 + c * 16;                                              \
            src = ok ? src : zp;                                                                           \
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(wbuf + prow * 128), 16, 0, 0);               \
        }                                                                                                  \
    }
    // window of the first iteration and the new rows of the next PF - 1, by all eight waves
    RW32_ROWS(0LL, 0, (R16 >> 3) + 16 * TPI * (PF - 1), wave, 8)
    {
        conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
        if (tid < COUT / 4) {
            sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + 4 * tid);
            bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + 4 * tid);
        }
        static_assert(conv_epi::EpiArea<COUT>::BYTES <= 2048, "epilogue area");
        conv_epi::prepare<COUT>(p, epi_area, 0, sc_pre, bs_pre);
    }
    // per-iteration advance of a grid position by STEP pixels: (sn, sy, sx) = STEP in (image, row, column) digits
    const unsigned W_ = (unsigned)p.W, H_ = (unsigned)p.H;
    const unsigned sx = (unsigned)STEP % W_, syf = (unsigned)STEP / W_;
    const unsigned sy = syf % H_, sn = syf / H_;
    // (n, y, x) of the first grid pixel this lane STORES in an iteration (row (lane >> 3) of its 32-pixel slice) and
    // (y, x) of the two pixels it COMPUTES (rows (lane & 15) and 16 + (lane & 15); PADDED only: tap validity)
    unsigned en, ey, ex, cy[2] = {0, 0}, cx[2] = {0, 0};
    {
        const unsigned hw = H_ * W_;
        const unsigned g0 = (unsigned)(POOL ? g_start : t_begin * 128) + (TPI == 2 ? grp * 128 : 0) + ms * 32;     // grid pixels < 2^31 (launcher)
        const unsigned ge = g0 + (lane >> 3);
        en = ge / hw;
        const unsigned rem = ge - en * hw;
        ey = rem / W_; ex = rem - ey * W_;
        if (PADDED || POOL) {
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const unsigned gc = g0 + pi * 16 + (lane & 15);
                const unsigned remc = gc % hw;
                cy[pi] = remc / W_; cx[pi] = remc - cy[pi] * W_;
            }
        }
    }
    // POOL: (n, y, x) of the tile's first grid pixel (wave-uniform), advanced with the tiles
    unsigned tn_ = 0, ty_ = 0, tx_ = 0;
    if (POOL) {
        const unsigned hw = H_ * W_;
        tn_ = (unsigned)g_start / hw;
        ty_ = ((unsigned)g_start - tn_ * hw) / W_;         // the walk starts at the beginning of a grid row: tx_ = 0
    }
    // tap offsets in ring rows (scalars) and this lane's window row at ring position 0
    int toff[ntaps];
#pragma unroll
    for (int t = 0; t < ntaps; ++t) toff[t] = (t / KWC - p.PH) * p.W + (t % KWC - p.PW);
    const int lrow0 = (TPI == 2 ? grp * 128 : 0) + ms * 32 + (lane & 15) - minoff;   // window row of this lane's first tile row at offset 0
    const int l4 = lane >> 4;
    // byte address (inside the ring) of this lane's pixel-half-0 hi fragment of every tap: advanced by STEP rows per iteration
    // and wrapped (STEP and the ring are multiples of 16 rows, which leave the chunk swizzle of a row unchanged), instead of
    // being rebuilt from the row number nine times per tile (14 -> 8 vector instructions per tap)
    const unsigned ringB = (unsigned)ring * 128u;
    unsigned a0t[ntaps];
#pragma unroll
    for (int t = 0; t < ntaps; ++t) {
        int wrow = lrow0 + toff[t];                        // window position 0; < 2 * ring
        wrow = wrow >= ring ? wrow - ring : wrow;
        a0t[t] = (unsigned)(wrow * 128 + ((l4 ^ tise_lds_swz(wrow)) << 4));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    conv_epi::Acc16 acc_main[1][1], acc_corr[1][1];
    int wstart = 0;                                        // physical row of this iteration's window row 0: (it * STEP) mod ring
    int pnew = (R16 + (PF - 1) * STEP) % ring;             // physical row of the rows DMA'd in this iteration
    long long relnew = (long long)R16 + (PF - 1) * STEP;
    for (int it = 0; it < nit; ++it) {
        // the new rows of iteration it + PF replace the oldest STEP rows (last read before the previous barrier)
        const bool ahead = it + PF < nit;
        if (ahead && !(DBG && (p.nseg & 0x100))) { RW32_ROWS(relnew, pnew, 16 * TPI, wave, 8) }
        const long long tile = t_begin + (long long)it * TPI + (TPI == 2 ? grp : 0);
        unsigned tapmask[2] = {0x1ffu, 0x1ffu};            // per-lane tap validity of its two pixels (PADDED)
        if (PADDED) {
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                tapmask[pi] = 0u;
#pragma unroll
                for (int t = 0; t < ntaps; ++t) {
                    const int yy = (int)cy[pi] + t / KWC - p.PH, xx = (int)cx[pi] + t % KWC - p.PW;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) tapmask[pi] |= 1u << t;
                }
            }
        }
        half8_t fa_[2][2][2];                              // [buffer][pixel half][hi / lo]
// the four fragments of a tap (16 pixels x 32 channels each): pixel half 0 hi at the swizzled chunk (lane >> 4), lo = chunk ^ 4;
// pixel half 1 sixteen ring rows further (same swizzle: the ring is a multiple of 16 rows), wrapped on its own
#define RW32_READS(TAP, BUF)                                                                              \
        {                                                                                                  \
            const unsigned a0 = a0t[TAP];                                                                  \
            const unsigned a1u = a0 + 2048u - ringB;          /* wraps below zero unless row + 16 >= ring */ \
            const unsigned a1 = a1u < a1u + ringB ? a1u : a1u + ringB;                                     \
            u32x4_t ah0 = *reinterpret_cast<const u32x4_t*>(wbuf + a0);                                    \
            u32x4_t al0 = *reinterpret_cast<const u32x4_t*>(wbuf + (a0 ^ 64));                             \
            u32x4_t ah1 = *reinterpret_cast<const u32x4_t*>(wbuf + a1);                                    \
            u32x4_t al1 = *reinterpret_cast<const u32x4_t*>(wbuf + (a1 ^ 64));                             \
            if (PADDED) {                                                                                  \
                const unsigned am0 = (tapmask[0] >> (TAP)) & 1u ? 0xffffffffu : 0u;                        \
                const unsigned am1 = (tapmask[1] >> (TAP)) & 1u ? 0xffffffffu : 0u;                        \
                ah0 &= am0; al0 &= am0; ah1 &= am1; al1 &= am1;                                            \
            }                                                                                              \
            fa_[BUF][0][0] = __builtin_bit_cast(half8_t, ah0);                                             \
            fa_[BUF][0][1] = __builtin_bit_cast(half8_t, al0);                                             \
            fa_[BUF][1][0] = __builtin_bit_cast(half8_t, ah1);                                             \
            fa_[BUF][1][1] = __builtin_bit_cast(half8_t, al1);                                             \
        }
#define RW32_MFMAS(TAP, BUF)                                                                              \
        _Pragma("unroll") for (int ci = 0; ci < 2; ++ci)                                                   \
            _Pragma("unroll") for (int pi = 0; pi < 2; ++pi) {                                             \
                float4_t& cm = acc_main[0][0].v[ci][pi];                                                   \
                float4_t& cc = acc_corr[0][0].v[ci][pi];                                                   \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[TAP][ci][1], fa_[BUF][pi][0], cc, 0, 0, 0); \
                cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[TAP][ci][0], fa_[BUF][pi][0], cm, 0, 0, 0); \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(bw[TAP][ci][0], fa_[BUF][pi][1], cc, 0, 0, 0); \
            }
// all nine taps of this wave's 32 pixels x 32 couts into (acc_main, acc_corr)
#define RW32_TAPS()                                                                                       \
        {                                                                                                  \
            conv_epi::acc_zero(acc_main[0][0]); conv_epi::acc_zero(acc_corr[0][0]);                        \
            if (!(DBG && (p.nseg & 0x200))) {                                                              \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            RW32_READS(0, 0)                                                                               \
            _Pragma("unroll") for (int tap = 0; tap < ntaps; ++tap) {                                      \
                if (tap + 1 < ntaps) {                                                                     \
                    if ((tap + 1) & 1) { RW32_READS(tap + 1, 1) } else { RW32_READS(tap + 1, 0) }          \
                }                                                                                          \
                if (tap & 1) { RW32_MFMAS(tap, 1) } else { RW32_MFMAS(tap, 0) }                            \
            }                                                                                              \
            /* issue order: the four fragment reads of tap t + 1, THEN the twelve MFMAs of tap t (left to itself the \
               scheduler put every read right in front of its MFMA and an lgkmcnt(0) between them: the LDS latency  \
               was exposed eighteen times per tile) */                                                     \
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                             \
            _Pragma("unroll") for (int tap = 0; tap + 1 < ntaps; ++tap) {                                  \
                __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                         \
                __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                        \
            }                                                                                              \
            __builtin_amdgcn_sched_group_barrier(0x008, 12, 0);                                            \
            __builtin_amdgcn_sched_barrier(0);                                                             \
            }                                                                                              \
        }
#define RW32_EPI(TILE, N_, Y_, X_)                                                                        \
        if (!(DBG && (p.nseg & 0x400))) {                                                                  \
            const unsigned nyx[3] = {N_, Y_, X_};                                                          \
            conv_epi::store_tiles_desc<1, 1, true, COUT>(p, acc_main, acc_corr, stage, epi_area, (TILE) * 128 + ms * 32, \
                                                         COUT == 64 ? grp * 4 : 0, nyx);                   \
        }
        const bool live = wrap_guard(tile, ntiles);
        // advance of the grid coordinates by STEP pixels
#define RW32_ADVANCE()                                                                                    \
        {                                                                                                  \
            ex += sx; ey += sy; en += sn;                                                                  \
            if (ex >= W_) { ex -= W_; ++ey; }                                                              \
            if (ey >= H_) { ey -= H_; ++en; }                                                              \
        }
        if (POOL || live) RW32_TAPS()
        // the rows of the next iteration have landed.  PF = 2: they were issued one iteration ago and only this
        // iteration's DMA is younger;  PF = 1: they are this iteration's DMA (issued before the taps above)
        if (PF == 2 && ahead) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if constexpr (POOL) {
            // ---- vertical running maximum, per lane (see the kernel header) --------------------------------------
            __syncthreads();                               // every wave has finished the horizontal pass of the previous iteration (HV is free)
            const unsigned grel0 = (unsigned)it * 128u + ms * 32 + (lane & 15);
            float4_t vold[2][2];
            unsigned va[2][2];
            bool valid[2], first[2];
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                const unsigned grel = grel0 + pi * 16;
                valid[pi] = cx[pi] < (unsigned)p.OW && cy[pi] < (unsigned)p.OH && grel < glen;
                first[pi] = grel < W_ || cy[pi] == 0;      // the workgroup's first conv row / an image's first: no window ends or continues here
                const int xs_ = valid[pi] ? (int)cx[pi] : 0;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int chunk = grp * 8 + ci * 4 + l4;
                    va[ci][pi] = (unsigned)(xs_ * 256 + ((chunk ^ pool_swz(xs_)) << 4));
                    vold[ci][pi] = *reinterpret_cast<const float4_t*>(vbuf + va[ci][pi]);
                }
            }
            float vmax = 0.f;
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int ch = grp * 32 + ci * 16 + l4 * 4;
                const float4_t sc = *reinterpret_cast<const float4_t*>(epi_area + conv_epi::EpiArea<COUT>::SCALE + ch * 4);
                const float4_t bs = *reinterpret_cast<const float4_t*>(epi_area + conv_epi::EpiArea<COUT>::BIAS + ch * 4);
#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    float4_t f, m;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = (acc_main[0][0].v[ci][pi][k] + acc_corr[0][0].v[ci][pi][k] * (1.0f / 2048.0f)) * sc[k];
                        f[k] = fmaxf(v + bs[k], 0.f);                           // RAW result: see the horizontal pass
                        vmax = fmaxf(vmax, f[k]);
                        m[k] = fmaxf(vold[ci][pi][k], f[k]);
                    }
                    const bool even = !(cy[pi] & 1u);
                    if (valid[pi]) {
                        float4_t nv;
#pragma unroll
                        for (int k = 0; k < 4; ++k) nv[k] = (first[pi] || even) ? f[k] : m[k];
                        *reinterpret_cast<float4_t*>(vbuf + va[ci][pi]) = nv;
                        if (even && !first[pi]) {
                            const int xx = (int)cx[pi];
                            const int slot = xx >= POOL_HV_SLOTS ? xx - POOL_HV_SLOTS : xx;
                            const int chunk = grp * 8 + ci * 4 + l4;
                            *reinterpret_cast<float4_t*>(hvbuf + slot * 256 + ((chunk ^ pool_swz(slot)) << 4)) = m;
                        }
                    }
                }
            }
            tise_flag_split_overflow(vmax);
            __syncthreads();                               // HV of this iteration complete
            // ---- horizontal pass: the windows whose last column (2 ox + 2) lies in this tile's even-row segment ----
            {
                const unsigned len1 = (W_ - tx_) < 128u ? (W_ - tx_) : 128u;            // pixels of the tile in grid row ty_
                const bool ev1 = !(ty_ & 1u);
                const unsigned ye = ev1 ? ty_ : ty_ + 1u;                               // the even one of the tile's (at most two) rows
                const int xs = ev1 ? (int)tx_ : 0;
                int xe = ev1 ? (int)(tx_ + len1) - 1 : 127 - (int)len1;                 // (odd first row filling the tile: xe = -1)
                const unsigned row_rel = (unsigned)it * 128u + (ev1 ? 0u : len1) - (unsigned)xs;   // of the row's first pixel, relative to the walk
                xe = xe < p.OW - 1 ? xe : p.OW - 1;
                const bool row_ok = ye >= 2u && ye < (unsigned)p.OH && row_rel != 0u && row_rel < glen && xe >= 2;
                if (row_ok) {
                    const int ox_lo = xs <= 2 ? 0 : (xs - 1) >> 1, ox_hi = (xe - 2) >> 1;
                    const int ox = ox_lo + (tid >> 3), c8 = tid & 7;
                    if (ox <= ox_hi) {
                        float mx[8];
#pragma unroll
                        for (int j = 0; j < 3; ++j) {
                            const int xx = 2 * ox + j;
                            const int slot = xx >= POOL_HV_SLOTS ? xx - POOL_HV_SLOTS : xx;
                            const unsigned a = (unsigned)(slot * 256 + (((2 * c8) ^ pool_swz(slot)) << 4));
                            const float4_t q0 = *reinterpret_cast<const float4_t*>(hvbuf + a);
                            const float4_t q1 = *reinterpret_cast<const float4_t*>(hvbuf + (a ^ 16u));
#pragma unroll
                            for (int k = 0; k < 4; ++k) {
                                mx[k] = j == 0 ? q0[k] : fmaxf(mx[k], q0[k]);
                                mx[4 + k] = j == 0 ? q1[k] : fmaxf(mx[4 + k], q1[k]);
                            }
                        }
                        // V / HV hold RAW fp32 results r.  F = merge(split(.)) is monotone, so max_i F(r_i) = F(max_i r_i): the
                        // maximum of the raw values is split, merged and split again -- exactly split(max_i merge(split(r_i))),
                        // what pooling the stored split tensor gives -- on a quarter as many values as a split + merge per result
                        half8_t ph, pl;
#pragma unroll
                        for (int k = 0; k < 8; ++k) {
                            const _Float16 h0 = (_Float16)mx[k];
                            const _Float16 l0 = (_Float16)((mx[k] - (float)h0) * 2048.0f);
                            const float fm = (float)h0 + (float)l0 * (1.0f / 2048.0f);
                            ph[k] = (_Float16)fm;
                            pl[k] = (_Float16)((fm - (float)ph[k]) * 2048.0f);
                        }
                        const conv_epi::ChunkDesc cd = *reinterpret_cast<const conv_epi::ChunkDesc*>(epi_area + conv_epi::EpiArea<COUT>::DESC + c8 * 32);
                        // the walk crossed into the next image when the even row is row 0 of it: ye >= 2 excludes that case
                        const unsigned long long pix = ((unsigned long long)tn_ * (unsigned)ohp + ((ye - 2u) >> 1)) * (unsigned)owp + (unsigned)ox;
                        unsigned char* d = reinterpret_cast<unsigned char*>(cd.base) + pix * (unsigned long long)cd.row_stride;
                        if (cd.valid) {
                            __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, ph), conv_epi::global_ptr(d));
                            __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, pl), conv_epi::global_ptr(d + cd.second));
                        }
                    }
                }
                tx_ += sx; ty_ += sy; tn_ += sn;
                if (tx_ >= W_) { tx_ -= W_; ++ty_; }
                if (ty_ >= H_) { ty_ -= H_; ++tn_; }
            }
        } else {
            if (live) RW32_EPI(tile, en, ey, ex)
        }
        RW32_ADVANCE()
        // next iteration: ring positions STEP rows further
        wstart += STEP; wstart = wstart >= ring ? wstart - ring : wstart;
#pragma unroll
        for (int t = 0; t < ntaps; ++t) {
            const unsigned u = a0t[t] + (unsigned)(STEP * 128) - ringB;
            a0t[t] = u < u + ringB ? u : u + ringB;        // unsigned: u wrapped below zero <=> no ring wrap
        }
        pnew += STEP; pnew = pnew >= ring ? pnew - ring : pnew;
        relnew += STEP;
        if (PADDED || POOL) {
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                cx[pi] += sx; cy[pi] += sy;
                if (cx[pi] >= W_) { cx[pi] -= W_; ++cy[pi]; }
                if (cy[pi] >= H_) cy[pi] -= H_;
            }
        }
        if (!POOL) __syncthreads();                        // (POOL: the barrier in front of the horizontal pass is the iteration's)
    }
#undef RW32_ADVANCE
#undef RW32_ROWS
#undef RW32_READS
#undef RW32_MFMAS
#undef RW32_TAPS
#undef RW32_EPI
}

template <int COUT, bool PADDED>
static int launch_regw32_inst(const ConvArgs* a, int R16, long long ntiles, long long grid, size_t lds, hipStream_t st) {
    static std::atomic<unsigned long long> attr_set{0};
    if (tise_first_use_on_this_device(attr_set))
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_regw32_kernel<COUT, PADDED>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((conv_regw32_kernel<COUT, PADDED>), dim3((unsigned)grid), dim3(512), lds, st, *a, R16, ntiles);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

// Conv2d_2b with MaxPool2d(3, 2) in the epilogue (POOL instance): unpadded 3x3 on a zero-bordered input, 64 couts, split
// destinations of the POOLED grid ((OH - 3) / 2 + 1) x ((OW - 3) / 2 + 1).
int launch_regw32_pool(const ConvArgs* a, hipStream_t st) {
    if (a->Cin != 32 || a->SH != 1 || a->SW != 1 || a->KH != 3 || a->KW != 3 || a->Kpad != 9 * 32 || a->Cout != 64 || (a->PH | a->PW) != 0 ||
        a->OH < 3 || a->OW < 3 || a->W < 128 || a->W > 2 * POOL_HV_SLOTS || (long long)a->N * a->H * a->W >= 0x7fffff00LL || a->out_hp != 0)
        return TISE_ERR_INVALID_ARG;
    for (int i = 0; i < (a->nseg & 0xff); ++i)
        if (a->seg[i].mode != 0) return TISE_ERR_INVALID_ARG;  // pooled raw fp32 segments: not provided
    if (a->nseg & ~0xff) return TISE_ERR_INVALID_ARG;
    const int R16 = (128 + 2 * a->W + 2 + 15) & ~15;
    const size_t lds = (size_t)(R16 + 256) * 128 + 2048 + (size_t)a->OW * 256 + (size_t)POOL_HV_SLOTS * 256;
    if (lds > 160 * 1024) return TISE_ERR_UNSUPPORTED;
    const int ncu_cached = tise_cu_count();                   // per device, atomic (common.h)
    const long long units = (long long)a->N * ((a->OH - 3) / 2 + 1);
    long long grid = (units + 3) / 4;                         // >= 4 pool rows per workgroup: one conv row per boundary is computed twice
    grid = grid < 1 ? 1 : (grid > ncu_cached ? ncu_cached : grid);
    static std::atomic<unsigned long long> attr_set{0};
    if (tise_first_use_on_this_device(attr_set))
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_regw32_kernel<64, false, false, true>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipLaunchKernelGGL((conv_regw32_kernel<64, false, false, true>), dim3((unsigned)grid), dim3(512), lds, st, *a, R16, 0x7fffffffffffLL);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int launch_regw32(const ConvArgs* a, hipStream_t st) {
    if (a->Cin != 32 || a->SH != 1 || a->SW != 1 || a->KH != 3 || a->KW != 3 || a->Kpad != 9 * 32 || a->W < 8 ||
        (a->Cout != 32 && a->Cout != 64) || (long long)a->N * a->H * a->W >= 0x7fffff00LL)
        return TISE_ERR_INVALID_ARG;
    const int tpi = 64 / a->Cout;
    const int R = 128 * tpi + 2 * a->W + 2;
    const int R16 = (R + 15) & ~15;
    const int pf = a->Cout == 64 ? 2 : 1;
    const size_t lds = (size_t)(R16 + 128 * tpi * pf) * 128 + 2048 + 8 * (size_t)conv_epi::Staging<1>::BYTES;
    if (lds > 160 * 1024) return TISE_ERR_UNSUPPORTED;
    const long long mg = (long long)a->N * a->H * a->W;
    const long long ntiles = (mg + 127) / 128;
    const int ncu_cached = tise_cu_count();                   // per device, atomic (common.h)
    const long long niter = (ntiles + tpi - 1) / tpi;
    const long long grid = niter < ncu_cached ? niter : ncu_cached;
    const bool padded = (a->PH | a->PW) != 0;
    if (a->nseg & 0x700) {                                     // tools/conv_ablate.py: the instrumented instances
        if (padded) return TISE_ERR_INVALID_ARG;
        hipLaunchKernelGGL((conv_regw32_kernel<64, false, true>), dim3((unsigned)grid), dim3(512), lds, st, *a, R16, ntiles);
        TISE_LAUNCH_CHECK();
        return a->Cout == 64 ? TISE_OK : TISE_ERR_INVALID_ARG;
    }
    if (a->Cout == 32)
        return padded ? launch_regw32_inst<32, true>(a, R16, ntiles, grid, lds, st) : launch_regw32_inst<32, false>(a, R16, ntiles, grid, lds, st);
    return padded ? launch_regw32_inst<64, true>(a, R16, ntiles, grid, lds, st) : launch_regw32_inst<64, false>(a, R16, ntiles, grid, lds, st);
}

// ------------------------------------------------------------------------------------------------
// Stem convolution Conv2d_1a_3x3 (3 -> 32 channels, 3x3, stride 2, valid; inception.py:60) on the matrix cores, from
// the Pillow-exact uint8 image of the resize kernel.  The fp32-FMA form (trunk_ops.hip) spends, per output pixel, 108
// table look-ups and 216 weight reads in LDS: it is bound by LDS operations (~80 per wave and 16 pixels), not by its
// 2.8 GB of output per 1000 images.  Here K = 27 taps padded to 32 is ONE K-step of the split-precision MFMA scheme of
// every other layer: the weights are 4 B-fragments in registers, and an input value costs one LDS gather -- the table
// holds the (hi, lo) fp16 pair of a byte's value as one dword -- and one v_perm per pair.
// K layout (chosen so that a lane's bytes sit at compile-time positions of two short runs): lane half h = lane >> 5
// of pixel (lane & 31) owns 16 slots u:  h = 0: u < 9 -> tap (kh 0, t = u), u >= 9 -> (kh 1, t = u - 9);  h = 1: u < 9 ->
// (kh 2, t = u), u = 9, 10 -> (kh 1, t = 7, 8), the rest zero weights;  t = 3 kw + cin;  MFMA k = 16 (u / 8) + 8 h + u % 8.
// The host packs the weights in that order (trunk.py: pack_stem_mfma).  Results are those of a split-precision layer
// (error ~1e-6 of the scale against fp64, like the other layers), not the bits of the fp32-FMA form.
__global__ __launch_bounds__(256) void stem_mfma_u8_kernel(const uint8_t* __restrict__ x, const float* __restrict__ lut,
                                                           int N, int H, int W, const _Float16* __restrict__ wsp,
                                                           const ConvArgs p) {
    constexpr int STG = conv_epi::Staging<1>::BYTES;
    __shared__ unsigned lut2[3 * 256];                       // (hi | lo << 16) of the network input value per channel and byte
    __shared__ __attribute__((aligned(16))) unsigned char epi[2048 + 4 * STG];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 3 * 256; i += 256) {
        const float v = lut[i];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * 2048.0f);
        lut2[i] = (unsigned)__builtin_bit_cast(unsigned short, hi) | ((unsigned)__builtin_bit_cast(unsigned short, lo) << 16);
    }
    {
        conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
        if (tid < 8) {
            sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + 4 * tid);
            bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + 4 * tid);
        }
        conv_epi::prepare<32>(p, epi, 0, sc_pre, bs_pre);
    }
    // weight fragments: cout = lane & 31, k = 16 s + 8 (lane >> 5) + 0..7; planes hi / lo of [32 couts][32 k]
    half8_t bw[2][2];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        bw[s2][0] = *reinterpret_cast<const half8_t*>(wsp + (lane & 31) * 32 + s2 * 16 + (lane >> 5) * 8);
        bw[s2][1] = *reinterpret_cast<const half8_t*>(wsp + 1024 + (lane & 31) * 32 + s2 * 16 + (lane >> 5) * 8);
    }
    __syncthreads();
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const unsigned M32 = (unsigned)p.M, ohw = (unsigned)(OH * OW);
    const unsigned ntiles = (M32 + 31u) / 32u;
    const int h = lane >> 5;
    const int row_bytes = W * 3;
    const uint8_t* x_end = x + (int64_t)N * H * W * 3;
    const unsigned lutb = (unsigned)(unsigned long long)(lds_ptr_t)lut2;
    unsigned char* stage = epi + 2048 + wave * STG;
// the two runs of this lane for tile T: REQUESTED one tile ahead of their use (raw aligned dwords + shift; the funnel
// shift that consumes them runs at the top of the next iteration): the chain load -> gather -> MFMA -> epilogue -> store
// of a tile is ~2 us of latency, and with the next tile's bytes already in flight the waves overlap it
#define STEM_FETCH(T, D1, S1, D2, S2)                                                                      \
    {                                                                                                      \
        const unsigned pix = (T) * 32u + (unsigned)(lane & 31);                                            \
        const unsigned pp = pix < M32 ? pix : 0u;             /* rows beyond M compute pixel 0 and are dropped by the epilogue */ \
        const unsigned n = pp / ohw;                                                                       \
        const unsigned rem = pp - n * ohw;                                                                 \
        const unsigned oh = rem / (unsigned)OW, ow = rem - oh * (unsigned)OW;                              \
        const uint8_t* base = x + (((int64_t)n * H + 2 * oh) * W + 2 * ow) * 3;                            \
        /* run 1: 9 bytes of row 0 (h = 0) / row 2 (h = 1);  run 2: row 1 from byte 0 (h = 0: 7 used) / byte 7 (h = 1: 2 used) */ \
        const uint8_t* q1 = base + (h ? 2 * row_bytes : 0);                                                \
        const uint8_t* q2 = base + row_bytes + (h ? 7 : 0);                                                \
        STEM_RUN(D1, S1, q1)                                                                               \
        STEM_RUN(D2, S2, q2)                                                                               \
    }
// four aligned dwords that contain the 12 bytes from Q, and the bit offset of Q inside them; the few runs whose aligned
// window would reach past the end of the tensor are read byte by byte (bytes past the end read as 0: their slots carry
// no weight)
#define STEM_RUN(D, S, Q)                                                                                  \
        {                                                                                                  \
            const uintptr_t a = reinterpret_cast<uintptr_t>(Q);                                            \
            const unsigned* w4 = reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3);                     \
            if (reinterpret_cast<const uint8_t*>(w4) + 16 <= x_end) {                                      \
                D[0] = w4[0]; D[1] = w4[1]; D[2] = w4[2]; D[3] = w4[3];                                    \
                S = (int)(a & 3) * 8;                                                                      \
            } else {                                                                                       \
                D[0] = D[1] = D[2] = D[3] = 0u;                                                            \
                for (int bb = 0; bb < 12; ++bb)                                                            \
                    if ((Q) + bb < x_end) D[bb >> 2] |= (unsigned)(Q)[bb] << (8 * (bb & 3));               \
                S = 0;                                                                                     \
            }                                                                                              \
        }
#define STEM_ALIGN(R, D, S)                                                                                \
        {                                                                                                  \
            const unsigned long long lo64 = ((unsigned long long)D[1] << 32) | D[0], mid64 = ((unsigned long long)D[2] << 32) | D[1], \
                                     hi64 = ((unsigned long long)D[3] << 32) | D[2];                       \
            R[0] = (unsigned)(lo64 >> S); R[1] = (unsigned)(mid64 >> S); R[2] = (unsigned)(hi64 >> S);     \
        }
    unsigned r1[3], r2[3], d1[4] = {0u, 0u, 0u, 0u}, d2[4] = {0u, 0u, 0u, 0u};
    int s1 = 0, s2s = 0;
    const unsigned tile0 = blockIdx.x * 4u + (unsigned)wave, tstep = gridDim.x * 4u;
    if (tile0 < ntiles) STEM_FETCH(tile0, d1, s1, d2, s2s)
    for (unsigned tile = tile0; tile < ntiles; tile += tstep) {
        STEM_ALIGN(r1, d1, s1)
        STEM_ALIGN(r2, d2, s2s)
        if (tile + tstep < ntiles) STEM_FETCH(tile + tstep, d1, s1, d2, s2s)
        // 16 slots -> LUT gathers: slot u < 9: run 1 byte u, channel u % 3;  u >= 9: run 2 byte b = u - 9, channel b % 3
        // (h = 0) or (b + 1) % 3 (h = 1: its run 2 starts at t = 7; only its first two slots carry weights)
        unsigned wv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int b = u < 9 ? u : u - 9;
            const unsigned word = (u < 9 ? r1 : r2)[b >> 2];
            const unsigned byte4 = ((word >> (8 * (b & 3))) & 0xffu) << 2;
            unsigned addr;
            if (u < 9) addr = lutb + (unsigned)((u % 3) * 1024) + byte4;
            else addr = lutb + byte4 + (h ? (unsigned)(((b + 1) % 3) * 1024) : (unsigned)((b % 3) * 1024));
            asm volatile("ds_read_b32 %0, %1" : "=v"(wv[u]) : "v"(addr));
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wv[0]), "+v"(wv[1]), "+v"(wv[2]), "+v"(wv[3]), "+v"(wv[4]), "+v"(wv[5]), "+v"(wv[6]), "+v"(wv[7]),
                     "+v"(wv[8]), "+v"(wv[9]), "+v"(wv[10]), "+v"(wv[11]), "+v"(wv[12]), "+v"(wv[13]), "+v"(wv[14]), "+v"(wv[15]));
        float16_t acc_main[1][1], acc_corr[1][1];
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc_main[0][0][j] = 0.f; acc_corr[0][0][j] = 0.f; }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            u32x4_t ah, al;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned w0 = wv[s2 * 8 + 2 * q], w1 = wv[s2 * 8 + 2 * q + 1];
                ah[q] = __builtin_amdgcn_perm(w1, w0, 0x05040100u);     // [w0.lo16, w1.lo16] = hi halves of slots 2q, 2q + 1
                al[q] = __builtin_amdgcn_perm(w1, w0, 0x07060302u);     // [w0.hi16, w1.hi16] = lo halves
            }
            const half8_t a_hi = __builtin_bit_cast(half8_t, ah), a_lo = __builtin_bit_cast(half8_t, al);
            acc_corr[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[s2][1], a_hi, acc_corr[0][0], 0, 0, 0);
            acc_main[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[s2][0], a_hi, acc_main[0][0], 0, 0, 0);
            acc_corr[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bw[s2][0], a_lo, acc_corr[0][0], 0, 0, 0);
        }
        conv_epi::store_tiles_desc<1, 1, false, 32>(p, acc_main, acc_corr, stage, epi, (long long)tile * 32);
    }
#undef STEM_FETCH
#undef STEM_RUN
#undef STEM_ALIGN
}

}  // namespace

extern "C" int tise_stem_conv3x3s2_split_u8_mfma(const uint8_t* x_dev, const float* lut_dev, int n, int h, int w, const void* wsplit_dev,
                                                 const float* scale_dev, const float* bias_dev, void* out_dev, void* stream) {
    if (!x_dev || !lut_dev || !wsplit_dev || !scale_dev || !bias_dev || !out_dev || n < 0 || h < 3 || w < 5) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const long long M = (long long)n * oh * ow;
    if (M >= 0x7fffff00LL || (long long)n * h * w * 3 < 32 || (reinterpret_cast<uintptr_t>(x_dev) & 3) != 0) return TISE_ERR_UNSUPPORTED;
    tise_conv_args a;
    memset(&a, 0, sizeof(a));
    a.scale = scale_dev; a.bias = bias_dev;
    a.N = n; a.H = h; a.W = w; a.Cin = 3; a.KH = 3; a.KW = 3; a.SH = 2; a.SW = 2; a.OH = oh; a.OW = ow;
    a.Cout = 32; a.K = 27; a.Kpad = 32; a.M = M; a.nseg = 1;
    a.seg[0].c0 = 0; a.seg[0].c1 = 32; a.seg[0].dst = out_dev; a.seg[0].ld = 32; a.seg[0].off = 0; a.seg[0].mode = 0;
    const long long tiles = (M + 31) / 32;
    long long grid = (tiles + 3) / 4;
    if (grid > 8192) grid = 8192;                              // the tables are built once per workgroup
    hipLaunchKernelGGL(stem_mfma_u8_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)stream, x_dev, lut_dev, n, h, w,
                       reinterpret_cast<const _Float16*>(wsplit_dev), a);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

// cfg 34: register-resident-weights sliding-window kernel (Cin = 32, 3x3, stride 1; Cout = 32 or 64, one launch);
// cfg 34 | 1024: the same with max_pool2d(3, 2) of the result taken in the epilogue (64 couts, unpadded).
int tise_conv_pipe_launch(const tise_conv_args* a, int cfg, void* stream) {
    if (a->out_hp && (a->out_y0 < 0 || a->out_x0 < 0 || a->out_y0 + a->OH > a->out_hp || a->out_x0 + a->OW > a->out_wp))
        return TISE_ERR_INVALID_ARG;
    {   // the grid epilogue follows the destination with 32-bit pixel indices and 32-bit signed byte increments
        const long long ohp = a->out_hp ? a->out_hp : a->OH, owp = a->out_hp ? a->out_wp : a->OW;
        long long ldmax = 0;
        for (int i = 0; i < (a->nseg & 0xff); ++i) ldmax = a->seg[i].ld > ldmax ? a->seg[i].ld : ldmax;
        if ((long long)a->N * ohp * owp >= 0xffffffffLL || (ohp + a->H) * owp * ldmax * 4 >= 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    }
    if (cfg == (34 | 1024)) return launch_regw32_pool(a, (hipStream_t)stream);
    if (cfg != 34) return TISE_ERR_INVALID_ARG;
    return launch_regw32(a, (hipStream_t)stream);
}

TISE_DEFINE_SPLIT_FLAG_READER(tise_internal_split_flag_conv_pipe)
