// Persistent, deep-pipelined form of the split-precision convolution (conv_split.hip explains the arithmetic:
// v ~= hi + lo * 2^-11, three fp16 MFMAs per product, fp32 accumulate; same operand formats, same results up
// to fp32 summation order).
//
// Why a second kernel: tools/conv_ablate.py on the 128-pixel, 2-stage kernels shows that neither the MFMAs
// (alone 55-65 % of the launch) nor the L2->LDS DMA (alone about the same) is saturated -- every K-step waits
// `vmcnt(0)` on a DMA issued one step earlier (~0.8 us round trip under load against ~0.6 us of MFMA work per
// CU-step), and the epilogue (15-40 % of a launch for K <= 1000) runs with the DMA engine idle.  That is the
// "two barriers per K-step" ceiling of cdna_hip_programming.md section 5 (~900 TF fp16; the trunk layers sit
// at 750-1000).  This kernel follows that section's way past it:
//   * ONE workgroup of 8 waves per CU, 256 pixels x BN couts per tile, all 160 KB of LDS: three operand stages
//     of 32 k each, the DMA (global_load_lds_dwordx4, 1 KB pieces) running TWO K-steps ahead of the MFMAs;
//   * a counted `s_waitcnt vmcnt(L)` (L = pieces one wave issues per stage) and a raw `s_barrier` per step:
//     the newest stage stays in flight across the barrier, nothing in the loop ever drains the queue;
//   * PERSISTENT: each workgroup walks tiles v = it * gridDim + slot and the DMA cursor runs ahead of the MFMA
//     cursor ACROSS tile boundaries, so the first two stages of the next tile land while this tile's epilogue
//     runs; the epilogue stages through the one LDS stage that is free at that moment, privately per wave
//     (no workgroup barrier inside it);
//   * the K loop is (tap, 32-channel block) with pointer increments only (as the "fast" kernel); Cin that is
//     not a multiple of 32 (80, 48) is handled by zero-padded weights [tap][Cin rounded up to 32] and a per-lane
//     channel guard on the pixel operand;
//   * the MFMA operands are swapped (weights first): the accumulator of a lane then holds 4 CONSECUTIVE couts
//     of one pixel per register quad, so the epilogue converts and stages with 8-byte LDS writes.
// Ordering rules (MI355X_MICROARCH.md, "Read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  stage of step g was issued during step g-2; each wave's vmcnt(L) at the top of step g retires its own
//        pieces of it (loads return in order; stores still in flight only make the wait longer, never shorter),
//        the barrier that follows makes every wave's pieces visible to every reader.
//   WAR  the DMA for step g+2 overwrites the stage read in step g-1; it is issued after the barrier of step g,
//        which a wave reaches only after the ds_reads feeding its step g-1 MFMAs have returned.
//   Epilogue: `vmcnt(0)` + barrier after the tile's last step (all fragment reads of the stage that becomes
//        the staging area are done; everything issued so far has landed, so the next two steps need no wait),
//        and the staging stage is not overwritten before the next step's barrier, which every wave reaches only
//        after its epilogue.
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

#define CP_BM 256
#define CP_BK 32
#define CP_TWAVE (conv_epi::Staging<1>::BYTES)   // staging bytes per wave (one 32 x 32 accumulator tile)

// WN waves along couts, 8 / WN along pixels; a wave owns TMW x TNW accumulator tiles of 32 x 32.
template <int WN, int TMW, int TNW, bool PP>
__global__ __launch_bounds__(512, 1) void conv_pipe_kernel(const ConvArgs p, const int ncb, const int tiles_n,
                                                           const long long ntiles) {
    constexpr int WM = 8 / WN;
    constexpr bool EPI_GRID = false;
    static_assert(WM * TMW * 32 == CP_BM, "tile is 256 pixels");
    constexpr int BN = 32 * TNW * WN;
    constexpr int A_PLANE = CP_BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int NBPIECE = BN / 8;                       // weight DMA pieces per stage (2 planes x BN/16)
    constexpr int NBP = (NBPIECE + 7) / 8;                // per wave (surplus slots repeat a piece)
    constexpr int L = 4 + NBP;                            // DMA pieces one wave issues per stage
    static_assert(8 * CP_TWAVE <= STAGE, "epilogue staging must fit one stage");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    // slot of this workgroup in the tile order: workgroups of one XCD (blockIdx % 8) take neighbouring tiles
    const long long G = (long long)gridDim.x;
    long long slot = blockIdx.x;
    {
        const long long q = G >> 3, r = G & 7, xcd = slot & 7, idx = slot >> 3;
        slot = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ntaps = p.KH * p.KW;
    const int nsteps = ntaps * ncb;                       // K-steps per tile
    const int kstride = nsteps * CP_BK;                   // halfs per weight row
    const long long my_tiles = slot < ntiles ? (ntiles - slot + G - 1) / G : 0;
    const long long total = my_tiles * nsteps;            // K-steps this workgroup computes
    if (total == 0) return;

    const int cl = (lane & 3) ^ ((lane >> 4) & 3);        // logical 16-byte chunk this lane's DMA pieces fetch
    const int cleft = p.Cin - cl * 8;                     // channel guard: block cb is real iff cb * 32 < cleft
    const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.w);
    const _Float16* zp = reinterpret_cast<const _Float16*>(g_pipe_zero_page);

    // ---- DMA cursor ---------------------------------------------------------------------------------
    long long is_v = slot;                                // tile being fetched
    int is_kh = 0, is_kw = 0, is_cb = 0, is_stage = 0;
    long long issued = 0;
    int ih0[2], iw0[2];
    const _Float16* img[2];
    const _Float16* pa[2];
    bool rok[2];
    const _Float16* pb[NBP];
    int pb_off[NBP];
    long long pb_row[NBP];
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int q = (wave + 8 * i) % NBPIECE;
        const int plane = q >= NBPIECE / 2 ? 1 : 0;
        const int rb = q - plane * (NBPIECE / 2);
        pb_row[i] = (plane ? p.w_plane : 0) + (long long)(rb * 16 + (lane >> 2)) * kstride + cl * 8;
        pb_off[i] = 2 * A_PLANE + plane * B_PLANE + rb * 1024;
    }
#define CP_TAP()                                                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                     \
        const int ih = ih0[jj] + is_kh, iw = iw0[jj] + is_kw;                                              \
        const bool ok = rok[jj] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;                             \
        pa[jj] = ok ? img[jj] + ((long long)ih * p.W + iw) * p.Cin : nullptr;                              \
    }
#define CP_TILE()                                                                                         \
    {                                                                                                     \
        const long long tm_ = is_v / tiles_n;                                                             \
        const int tn_ = (int)(is_v - tm_ * tiles_n);                                                      \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                 \
            const long long pix = tm_ * CP_BM + (2 * wave + jj) * 16 + (lane >> 2);                       \
            rok[jj] = pix < p.M;                                                                           \
            const long long pp = rok[jj] ? pix : 0;                                                        \
            const int ohw = p.OH * p.OW;                                                                   \
            const int n = (int)(pp / ohw);                                                                 \
            const int rem = (int)(pp - (long long)n * ohw);                                                \
            const int oh = rem / p.OW, ow = rem - oh * p.OW;                                               \
            ih0[jj] = oh * p.SH - p.PH;                                                                    \
            iw0[jj] = ow * p.SW - p.PW;                                                                    \
            img[jj] = xg + (long long)n * p.H * p.W * p.Cin + cl * 8;                                      \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NBP; ++i)                                                    \
            pb[i] = wgt + (long long)tn_ * BN * kstride + pb_row[i];                                       \
        CP_TAP()                                                                                           \
    }
#define CP_ISSUE()                                                                                        \
    {                                                                                                     \
        unsigned char* sb_ = lds + is_stage * STAGE;                                                       \
        const bool cok = is_cb * CP_BK < cleft;                                                            \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                 \
            const _Float16* pj = pa[jj];                                                                   \
            const bool ok = pj != nullptr && cok;                                                          \
            const _Float16* sh = ok ? pj : zp;                                                             \
            const _Float16* sl = ok ? pj + p.x_plane : zp;                                                 \
            unsigned char* da_ = sb_ + (2 * wave + jj) * 1024;                                             \
            __builtin_amdgcn_global_load_lds(sh, (lds_ptr_t)da_, 16, 0, 0);                                \
            __builtin_amdgcn_global_load_lds(sl, (lds_ptr_t)(da_ + A_PLANE), 16, 0, 0);                    \
            pa[jj] = pj ? pj + CP_BK : nullptr;                                                            \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < NBP; ++i) {                                                  \
            /* locals on purpose (array elements as direct builtin arguments make hipcc drop the host stub) */ \
            const _Float16* sw_ = pb[i];                                                                   \
            unsigned char* dw_ = sb_ + pb_off[i];                                                          \
            __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)dw_, 16, 0, 0);                               \
            pb[i] = sw_ + CP_BK;                                                                           \
        }                                                                                                  \
        is_stage = is_stage == 2 ? 0 : is_stage + 1;                                                       \
        ++issued;                                                                                          \
        if (++is_cb == ncb) {                                   /* wave-uniform: next tap / next tile */   \
            is_cb = 0;                                                                                     \
            if (++is_kw == p.KW) { is_kw = 0; ++is_kh; }                                                   \
            if (is_kh == p.KH) {                                                                           \
                is_kh = 0;                                                                                 \
                is_v += G;                                                                                 \
                if (issued < total) CP_TILE()                                                              \
            } else {                                                                                       \
                CP_TAP()                                                                                   \
            }                                                                                              \
        }                                                                                                  \
    }

    // ---- MFMA cursor --------------------------------------------------------------------------------
    float16_t acc_main[TMW][TNW], acc_corr[TMW][TNW];
#define CP_ZERO()                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                        \
        _Pragma("unroll") for (int t = 0; t < TNW; ++t)                                                    \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) { acc_main[i][t][j] = 0.f; acc_corr[i][t][j] = 0.f; }
    CP_ZERO()

    // fragment read offsets: row (lane & 31), logical chunk 2*s + (lane >> 5), swizzled with (row >> 2) & 3
    const int fswz = ((lane & 31) >> 2) & 3;
    const int fo0 = (lane & 31) * 64 + (((lane >> 5)) ^ fswz) * 16;
    const int fo1 = (lane & 31) * 64 + ((2 + (lane >> 5)) ^ fswz) * 16;
    const int fa_off = wm * TMW * 32 * 64;
    const int fb_off = 2 * A_PLANE + wn * TNW * 32 * 64;

    // One K-step: all fragment reads written first, the MFMAs after them, and a sched_group_barrier sequence
    // that makes the scheduler emit "2 reads, 3 MFMAs" alternately after the first 4 reads.
    constexpr int NREADS = 4 * (TMW + TNW), NGROUPS = 2 * TMW * TNW;
    constexpr int NR2 = (NREADS - 4) / 2;
#define CP_COMPUTE(SB)                                                                                    \
    {                                                                                                     \
        half8_t fa_[2][TMW][2], fb_[2][TNW][2];                                                            \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                    \
            const int fo = s ? fo1 : fo0;                                                                  \
            _Pragma("unroll") for (int i = 0; i < TMW; ++i) {                                              \
                const unsigned char* ap = (SB) + fa_off + i * 32 * 64 + fo;                                \
                fa_[s][i][0] = *reinterpret_cast<const half8_t*>(ap);                                      \
                fa_[s][i][1] = *reinterpret_cast<const half8_t*>(ap + A_PLANE);                            \
            }                                                                                              \
            _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                              \
                const unsigned char* bp = (SB) + fb_off + t * 32 * 64 + fo;                                \
                fb_[s][t][0] = *reinterpret_cast<const half8_t*>(bp);                                      \
                fb_[s][t][1] = *reinterpret_cast<const half8_t*>(bp + B_PLANE);                            \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                      \
            _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                \
                _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                          \
                    acc_main[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][i][0], acc_main[i][t], 0, 0, 0); \
                    acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][1], fa_[s][i][0], acc_corr[i][t], 0, 0, 0); \
                    acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][i][1], acc_corr[i][t], 0, 0, 0); \
                }                                                                                          \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                                 \
        _Pragma("unroll") for (int q = 0; q < NGROUPS; ++q) {                                              \
            if (q < NR2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                             \
        }                                                                                                  \
    }

    // ping-pong schedule: the fragment reads of a step happen one phase before its MFMAs
    half8_t pa_[2][TMW][2], pb_[2][TNW][2];
#define CP_READS(SB)                                                                                      \
    _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                        \
        const int fo = s ? fo1 : fo0;                                                                      \
        _Pragma("unroll") for (int i = 0; i < TMW; ++i) {                                                  \
            const unsigned char* ap = (SB) + fa_off + i * 32 * 64 + fo;                                    \
            pa_[s][i][0] = *reinterpret_cast<const half8_t*>(ap);                                          \
            pa_[s][i][1] = *reinterpret_cast<const half8_t*>(ap + A_PLANE);                                \
        }                                                                                                  \
        _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                                  \
            const unsigned char* bp = (SB) + fb_off + t * 32 * 64 + fo;                                    \
            pb_[s][t][0] = *reinterpret_cast<const half8_t*>(bp);                                          \
            pb_[s][t][1] = *reinterpret_cast<const half8_t*>(bp + B_PLANE);                                \
        }                                                                                                  \
    }
#define CP_MFMAS()                                                                                        \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                          \
        _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                    \
            _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                              \
                acc_main[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb_[s][t][0], pa_[s][i][0], acc_main[i][t], 0, 0, 0); \
                acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb_[s][t][1], pa_[s][i][0], acc_corr[i][t], 0, 0, 0); \
                acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pb_[s][t][0], pa_[s][i][1], acc_corr[i][t], 0, 0, 0); \
            }

    // ---- epilogue of the tile c_v, staging through the stage at SB (private 4.5 KB per wave): conv_epilogue.h
#define CP_EPILOGUE(SB)                                                                                   \
    {                                                                                                     \
        const long long tm_ = c_v / tiles_n;                                                              \
        const int n0 = (int)(c_v - tm_ * tiles_n) * BN;                                                   \
        const long long m0 = tm_ * CP_BM;                                                                 \
        conv_epi::store_tiles<TMW, TNW, EPI_GRID, 1>(p, acc_main, acc_corr, (SB) + wave * CP_TWAVE,       \
                                                     m0 + wm * TMW * 32, n0 + wn * TNW * 32);             \
    }

    // ---- pipeline -----------------------------------------------------------------------------------
    // ablation switches for tools/conv_ablate.py (never set by the product path): 0x100 no DMA after the
    // prologue, 0x200 no fragment reads / MFMAs, 0x400 no epilogue
    const bool ab_dma = !(p.nseg & 0x100), ab_mma = !(p.nseg & 0x200), ab_epi = !(p.nseg & 0x400);
    CP_TILE()
    CP_ISSUE()
    if (issued < total) CP_ISSUE()
    long long c_v = slot;
    long long landed = 0;
    int c_step = 0, c_stage = 0;
    // in-kernel stamps (tools/conv_stamps.py, flag 0x800): waves 0 and 4 of workgroup 0 record s_memtime at five
    // points of each of the first 96 steps into 4 KB of LDS behind the stages; dumped to seg[3].dst at the end
    const bool stamps = (p.nseg & 0x800) && blockIdx.x == 0 && (wave & 3) == 0 && lane == 0;
    unsigned* stamp_lds = reinterpret_cast<unsigned*>(lds + 3 * STAGE) + (wave >> 2) * 512;
#define CP_STAMP(K)                                                                                       \
    if (stamps && done < 96) {                                                                             \
        stamp_lds[done * 5 + (K)] = (unsigned)__builtin_readcyclecounter();                                \
        if ((K) == 0 && (done == 20 || done == 60)) /* 100 MHz wall clock: calibrates the cycle counter */ \
            stamp_lds[480 + (done == 60)] = (unsigned)__builtin_amdgcn_s_memrealtime();                    \
    }
    if (!PP) {
        for (long long done = 0; done < total; ++done) {
            CP_STAMP(0)
            if (done >= landed) {
                if (issued > done + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            CP_STAMP(1)
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            CP_STAMP(2)
            if (issued < total) {
                if (ab_dma) CP_ISSUE() else ++issued;
            }
            CP_STAMP(3)
            const unsigned char* sb = lds + c_stage * STAGE;
            if (ab_mma) CP_COMPUTE(sb)
            CP_STAMP(4)
            if (++c_step == nsteps) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                landed = issued;
                unsigned char* st = lds + c_stage * STAGE;
                if (ab_epi || p.M < 0) CP_EPILOGUE(st)
                CP_ZERO()
                c_step = 0;
                c_v += G;
            }
            c_stage = c_stage == 2 ? 0 : c_stage + 1;
        }
    } else {
        // Ping-pong: waves 0-3 (group A) and 4-7 (group B) -- one of each per SIMD -- run the same two-phase step
        //   P1: fragment reads of step g -> registers, DMA of step g+2, lgkmcnt(0)     P2: the step's MFMAs
        // one barrier apart, so that on every SIMD one wave is in its MFMA phase while the other issues DMA and
        // LDS reads (the DMA issue blocks its wave for 100-400 cycles per piece once the TA queue fills: in-kernel
        // stamps, tools/conv_stamps.py).  Barrier #2g ends B's P2(g-1) and A's P2(g-1)+1 ... ordering:
        //   stage g is read by A in interval 2g and by B in interval 2g+1, so every wave retires its own pieces of
        //   stage g+1 (counted vmcnt, the newest stage stays in flight) before barrier #2g+2: A at the end of its
        //   P2(g), B at the end of its P1(g); the DMA of step g+2 overwrites stage g-1, whose last reads (B's
        //   P1(g-1)) returned before barrier #2g.
        const bool grpB = wave >= 4;
        long long done = 0;
        if (issued > 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        landed = 1;
#define CP_CERTIFY()                                                                                      \
        if (done + 1 < total && done + 1 >= landed) {                                                      \
            if (issued > done + 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");                \
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                          \
        }
        for (long long tile = 0; tile < my_tiles; ++tile) {
            if (grpB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            for (int step = 0; step < nsteps; ++step) {
                CP_STAMP(0)
                const unsigned char* sb = lds + c_stage * STAGE;
                if (ab_mma) { CP_READS(sb) }
                if (issued < total) {
                    if (ab_dma) CP_ISSUE() else ++issued;
                }
                CP_STAMP(1)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (grpB) { CP_CERTIFY() }
                CP_STAMP(2)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                CP_STAMP(3)
                __builtin_amdgcn_s_setprio(1);
                if (ab_mma) { CP_MFMAS() }
                __builtin_amdgcn_s_setprio(0);
                if (!grpB) { CP_CERTIFY() }
                CP_STAMP(4)
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ++done;
                c_stage = c_stage == 2 ? 0 : c_stage + 1;
            }
            if (!grpB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            landed = issued;
            unsigned char* st = lds + (c_stage == 0 ? 2 : c_stage - 1) * STAGE;
            if (ab_epi || p.M < 0) CP_EPILOGUE(st)
            CP_ZERO()
            c_v += G;
            __builtin_amdgcn_s_barrier();            // staging area free before anyone's next DMA lands in it
            asm volatile("" ::: "memory");
        }
    }
    if (stamps) {
        unsigned* out = reinterpret_cast<unsigned*>(p.seg[3].dst) + (wave >> 2) * 512;
        for (int i = 0; i < 482; ++i) out[i] = stamp_lds[i];
    }
}

template <int WN, int TMW, int TNW, bool PP>
int launch_cfg(const ConvArgs* a, hipStream_t st) {
    constexpr int BN = 32 * TNW * WN;
    constexpr int STAGE = 2 * CP_BM * 64 + 2 * BN * 64;
    constexpr int LDS = 3 * STAGE + ((3 * STAGE + 4096 <= 160 * 1024) ? 4096 : 0);   // + stamp area when it fits
    if ((a->nseg & 0x800) && LDS == 3 * STAGE) return TISE_ERR_UNSUPPORTED;
    static bool attr_set = false;
    if (!attr_set) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipe_kernel<WN, TMW, TNW, PP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int ncb = (a->Cin + CP_BK - 1) / CP_BK;
    const int tiles_n = (a->Cout + BN - 1) / BN;
    const long long ntiles = ((a->M + CP_BM - 1) / CP_BM) * tiles_n;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        TISE_HIP_CHECK(hipGetDevice(&dev));
        TISE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL((conv_pipe_kernel<WN, TMW, TNW, PP>), dim3((unsigned)grid), dim3(512), LDS, st, *a, ncb, tiles_n, ntiles);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}


// ------------------------------------------------------------------------------------------------
// Window form (stride-1 convolutions with more than one tap).  glds_rate (tools/microbench) puts the LDS-DMA
// path at 30 B/clk/CU for the 16-row x 64-B pieces used here (38 for 128-B rows): a 256 x 128 x 32 step needs
// 48 KB = ~1500 DMA cycles against 1536 MFMA cycles per SIMD, so the per-tap implicit GEMM is DMA-bound
// whatever the schedule.  Here the 256 tile rows are consecutive pixels of the INPUT grid and a tap is a row
// offset (kh-PH)*W + (kw-PW) into one resident WINDOW of 256 + (KH-1)*W + KW-1 grid pixels per 32-channel
// block: the window is fetched once per channel block (double-buffered, prefetched during the previous
// block's taps), only the weight tile (BN x 32 k, 3 stages, two steps ahead, counted vmcnt) streams per step:
// ~22 KB instead of 48 KB per step for a 3x3 at 35^2.  Border handling as conv_split.hip's window variant:
// valid convolutions compute the grid pixels without an output and drop them in the epilogue; padded ones
// mask, per lane and tap, fragments whose source lies outside the image.
// K order is (channel block, tap); weights packed [cout][tap][Cin rounded up to 32] as for the kernel above.
template <int WN, int TMW, int TNW, bool PP>
__global__ __launch_bounds__(512, 1) void conv_pipew_kernel(const ConvArgs p, const int ncb, const int tiles_n,
                                                            const int R16, const int nwin) {
    constexpr int WM = 8 / WN;
    constexpr bool EPI_GRID = true;
    static_assert(WM * TMW * 32 == CP_BM, "tile is 256 pixels");
    constexpr int BN = 32 * TNW * WN;
    constexpr int B_PLANE = BN * 64;
    constexpr int BSTAGE = 2 * B_PLANE;
    constexpr int NBPIECE = BN / 8;
    constexpr int NBP = (NBPIECE + 7) / 8;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int win_plane = R16 * 64;
    const int win_bytes = 2 * win_plane;
    unsigned char* bst = lds + nwin * win_bytes;          // three weight stages behind the window buffer(s)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave - wm * WN;
    const long long nwg = (long long)gridDim.x;
    long long c_v = blockIdx.x;
    {
        const long long q = nwg >> 3, r = nwg & 7, xcd = c_v & 7, idx = c_v >> 3;
        c_v = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long tile_m = c_v / tiles_n;
    const int n0t = (int)(c_v - tile_m * tiles_n) * BN;
    const long long m0t = tile_m * CP_BM;
    const long long mgrid = (long long)p.N * p.H * p.W;
    const int ntaps = p.KH * p.KW;
    const int nsteps = ncb * ntaps;
    const int kstride = nsteps * CP_BK;
    const int minoff = -p.PH * p.W - p.PW;
    const int wpieces = 2 * (R16 >> 4);                   // DMA pieces of one window (2 planes)
    const int wpw = (wpieces + 7) >> 3;                   // per wave (surplus slots repeat a piece)

    const int cl = (lane & 3) ^ ((lane >> 4) & 3);
    const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.w);
    const _Float16* zp = reinterpret_cast<const _Float16*>(g_pipe_zero_page);

    // per-lane tap validity (padded convolutions), one mask per accumulator row tile
    unsigned tapmask[TMW];
#pragma unroll
    for (int i = 0; i < TMW; ++i) {
        tapmask[i] = 0xffffffffu;
        if (p.PH | p.PW) {
            const long long g = m0t + (wm * TMW + i) * 32 + (lane & 31);
            const long long hw = (long long)p.H * p.W;
            const int rem = (int)(g % hw);
            const int y = rem / p.W, x = rem - y * p.W;
            unsigned m = 0u;
            for (int kh = 0, t = 0; kh < p.KH; ++kh)
                for (int kw = 0; kw < p.KW; ++kw, ++t) {
                    const int yy = y + kh - p.PH, xx = x + kw - p.PW;
                    if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) m |= 1u << t;
                }
            tapmask[i] = m;
        }
    }

    // weight DMA pointers (advance 32 halfs per step: K order (cb, tap) is the packed order [tap][cb] permuted,
    // so the pointer is rebuilt from (cb, tap) each step: one multiply-add)
    long long pb_row[NBP];
    int pb_off[NBP];
#pragma unroll
    for (int i = 0; i < NBP; ++i) {
        const int q = (wave + 8 * i) % NBPIECE;
        const int plane = q >= NBPIECE / 2 ? 1 : 0;
        const int rb = q - plane * (NBPIECE / 2);
        pb_row[i] = (plane ? p.w_plane : 0) + (long long)(n0t + rb * 16 + (lane >> 2)) * kstride + cl * 8;
        pb_off[i] = plane * B_PLANE + rb * 1024;
    }
#define CW_WEIGHTS(CB, TAP, SBUF)                                                                         \
    _Pragma("unroll") for (int i = 0; i < NBP; ++i) {                                                      \
        const _Float16* sw_ = wgt + pb_row[i] + ((TAP) * ncb + (CB)) * CP_BK;                              \
        unsigned char* dw_ = bst + (SBUF) * BSTAGE + pb_off[i];                                            \
        __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)dw_, 16, 0, 0);                                   \
    }
#define CW_WINDOW(CB, WBUF)                                                                               \
    {                                                                                                     \
        const int c = (CB) * CP_BK + cl * 8;                                                               \
        const bool c_ok = c < p.Cin;                                                                       \
        for (int k = 0; k < wpw; ++k) {                                                                    \
            int q = wave + 8 * k;                                                                          \
            q = q >= wpieces ? q - wpieces : q;                                                            \
            const int plane = q >= (R16 >> 4) ? 1 : 0;                                                     \
            const int rb = q - plane * (R16 >> 4);                                                         \
            const long long g = m0t + minoff + rb * 16 + (lane >> 2);                                      \
            const bool ok = c_ok && g >= 0 && g < mgrid;                                                   \
            const _Float16* src = xg + (plane ? p.x_plane : 0) + g * p.Cin + c;                            \
            src = ok ? src : zp;                                                                           \
            unsigned char* dst = lds + (WBUF) * win_bytes + plane * win_plane + rb * 1024;                 \
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);                               \
        }                                                                                                  \
    }
// wait until at most NB + EXTRA of this wave's DMA pieces are outstanding (immediate operand: dispatch on EXTRA)
#define CW_WAIT(EXTRA)                                                                                    \
    switch (EXTRA) {                                                                                      \
        case 0: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory"); break;                          \
        case 1: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 1) : "memory"); break;                      \
        case 2: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 2) : "memory"); break;                      \
        case 3: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 3) : "memory"); break;                      \
        case 4: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 4) : "memory"); break;                      \
        case 5: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 5) : "memory"); break;                      \
        case 6: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 6) : "memory"); break;                      \
        case 7: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 7) : "memory"); break;                      \
        case 8: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP + 8) : "memory"); break;                      \
        default: asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory"); break;                         \
    }

    float16_t acc_main[TMW][TNW], acc_corr[TMW][TNW];
    CP_ZERO()

    const int fswz = ((lane & 31) >> 2) & 3;
    const int fb0 = (lane & 31) * 64 + ((lane >> 5) ^ fswz) * 16;
    const int fb1 = (lane & 31) * 64 + ((2 + (lane >> 5)) ^ fswz) * 16;
    const int fb_off = wn * TNW * 32 * 64;
    const int lrow0 = wm * TMW * 32 + (lane & 31) - minoff;   // window row of accumulator row tile 0 at offset 0
    constexpr int NREADS = 4 * (TMW + TNW), NGROUPS = 2 * TMW * TNW;
    constexpr int NR2 = (NREADS - 4) / 2;

    // prologue: window of block 0, weights of steps 0 and 1
    CW_WINDOW(0, 0)
    CW_WEIGHTS(0, 0, 0)
    {
        const int t1 = ntaps > 1 ? 1 : 0, c1 = ntaps > 1 ? 0 : 1;
        if (nsteps > 1) CW_WEIGHTS(c1, t1, 1)
    }
    int cb = 0, tap = 0, kh = 0, kw = 0;
    int n_cb = ntaps > 2 ? 0 : (ntaps == 2 ? 1 : 2), n_tap = ntaps > 2 ? 2 : 0;   // (cb, tap) of step g+2
    if (ntaps == 1) { n_cb = 2; n_tap = 0; }
    int extra = 0;                                         // window pieces issued after the newest weights
    int stage = 0;
    half8_t fa_[2][TMW][2], fb_[2][TNW][2];
// fragments of the current tap: window row = tile row + tap offset
#define CW_READS()                                                                                        \
    {                                                                                                     \
        const int toff = (kh - p.PH) * p.W + (kw - p.PW);                                                  \
        const unsigned char* wbase = lds + (nwin > 1 ? (cb & 1) : 0) * win_bytes;                          \
        const unsigned char* bb = bst + stage * BSTAGE + fb_off;                                           \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                    \
            _Pragma("unroll") for (int i = 0; i < TMW; ++i) {                                              \
                const int wrow = lrow0 + i * 32 + toff;                                                    \
                const int ao = wrow * 64 + (((2 * s + (lane >> 5)) ^ ((wrow >> 2) & 3)) * 16);             \
                u32x4_t ah = *reinterpret_cast<const u32x4_t*>(wbase + ao);                                \
                u32x4_t al = *reinterpret_cast<const u32x4_t*>(wbase + win_plane + ao);                    \
                const unsigned am = (tapmask[i] >> tap) & 1u ? 0xffffffffu : 0u;                           \
                ah &= am; al &= am;                                                                        \
                fa_[s][i][0] = __builtin_bit_cast(half8_t, ah);                                            \
                fa_[s][i][1] = __builtin_bit_cast(half8_t, al);                                            \
            }                                                                                              \
            _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                              \
                const unsigned char* bp = bb + t * 32 * 64 + (s ? fb1 : fb0);                              \
                fb_[s][t][0] = *reinterpret_cast<const half8_t*>(bp);                                      \
                fb_[s][t][1] = *reinterpret_cast<const half8_t*>(bp + B_PLANE);                            \
            }                                                                                              \
        }                                                                                                  \
    }
#define CW_MFMAS()                                                                                        \
    _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                          \
        _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                    \
            _Pragma("unroll") for (int t = 0; t < TNW; ++t) {                                              \
                acc_main[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][i][0], acc_main[i][t], 0, 0, 0); \
                acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][1], fa_[s][i][0], acc_corr[i][t], 0, 0, 0); \
                acc_corr[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][i][1], acc_corr[i][t], 0, 0, 0); \
            }
// DMA of step g+2 (weights) and, at the first tap of a block, of the next block's window (after the weights)
#define CW_ISSUE()                                                                                        \
    {                                                                                                     \
        if (step + 2 < nsteps) {                                                                           \
            const int s2 = stage >= 1 ? stage - 1 : 2;     /* (stage + 2) % 3 */                           \
            CW_WEIGHTS(n_cb, n_tap, s2)                                                                    \
            if (++n_tap == ntaps) { n_tap = 0; ++n_cb; }                                                   \
        }                                                                                                  \
        if (tap == 0 && cb + 1 < ncb && nwin > 1) { CW_WINDOW(cb + 1, (cb + 1) & 1) }                      \
    }
#define CW_ADVANCE()                                                                                      \
    stage = stage == 2 ? 0 : stage + 1;                                                                    \
    if (++tap == ntaps) { tap = 0; kh = 0; kw = 0; ++cb; }                                                 \
    else if (++kw == p.KW) { kw = 0; ++kh; }
    if (!PP || nwin == 1 || ntaps < 3) {
        for (int step = 0; step < nsteps; ++step) {
            if (step + 1 < nsteps) { CW_WAIT(extra) } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            extra = (tap == 0 && cb + 1 < ncb && nwin > 1) ? wpw : 0;
            CW_ISSUE()
            CW_READS()
            CW_MFMAS()
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
            for (int q = 0; q < NGROUPS; ++q) {
                if (q < NR2) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
            }
            CW_ADVANCE()
            if (tap == 0 && nwin == 1 && cb < ncb) {       // single window buffer: reload it between channel blocks
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                CW_WINDOW(cb, 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
    } else {
        // Ping-pong schedule (see conv_pipe_kernel): waves 0-3 and 4-7 one barrier apart, P1 = fragment reads +
        // DMA issue, P2 = MFMAs.  A wave retires its pieces of the weights of step g+1 before barrier #2g+2 (group
        // A after its P2(g), group B after its P1(g)); pieces issued after them stay in flight: the weights of
        // step g+2 and a window issued in this or the previous step (ntaps >= 3: never both).
        const bool grpB = wave >= 4;
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBP) : "memory");      // window 0 and weights 0
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (grpB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
        bool prev_win = false;
        for (int step = 0; step < nsteps; ++step) {
            const bool this_win = tap == 0 && cb + 1 < ncb;
            CW_READS()
            CW_ISSUE()
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            const int ex = (this_win || prev_win) ? wpw : 0;
            if (grpB && step + 1 < nsteps) {
                if (step + 2 < nsteps) { CW_WAIT(ex) } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            __builtin_amdgcn_s_setprio(1);
            CW_MFMAS()
            __builtin_amdgcn_s_setprio(0);
            if (!grpB && step + 1 < nsteps) {
                if (step + 2 < nsteps) { CW_WAIT(ex) } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            prev_win = this_win;
            CW_ADVANCE()
        }
        if (!grpB) { __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    CP_EPILOGUE(lds)
}

template <int WN, int TMW, int TNW, bool PP>
int launch_win(const ConvArgs* a, hipStream_t st) {
    constexpr int BN = 32 * TNW * WN;
    const int ncb = (a->Cin + CP_BK - 1) / CP_BK;
    const int R = CP_BM + (a->KH - 1) * a->W + (a->KW - 1);
    const int R16 = (R + 15) & ~15;
    const size_t bst = 3 * (size_t)BN * 128;
    int nwin = ncb > 1 ? 2 : 1;
    if ((size_t)nwin * R16 * 128 + bst > 160 * 1024) nwin = 1;
    size_t lds = (size_t)nwin * R16 * 128 + bst;
    if (lds < 8 * CP_TWAVE) lds = 8 * CP_TWAVE;
    if (lds > 160 * 1024 || a->KH * a->KW > 32 || (nwin > 1 && (2 * (R16 >> 4) + 7) / 8 > 8)) return TISE_ERR_UNSUPPORTED;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipew_kernel<WN, TMW, TNW, PP>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const int tiles_n = (a->Cout + BN - 1) / BN;
    const long long mg = (long long)a->N * a->H * a->W;
    const long long ntiles = ((mg + CP_BM - 1) / CP_BM) * tiles_n;
    if (ntiles > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL((conv_pipew_kernel<WN, TMW, TNW, PP>), dim3((unsigned)ntiles), dim3(512), lds, st, *a, ncb, tiles_n, R16, nwin);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}


// ------------------------------------------------------------------------------------------------
// Resident-weights sliding-window kernel for the two 32-channel 3x3 layers of the stem (Conv2d_2a 149^2 x 32 -> 32
// valid, Conv2d_2b 147^2 x 32 -> 64 padded, as two 32-cout launches; 3.0 of the trunk's 21.5 conv ms).  With
// Cin = 32 a K-step is one tap, 6 MFMAs per wave per 32 couts, and the per-tap implicit GEMM moves 16 KB of pixels
// + 4 KB of weights for it: 5-6x more DMA cycles than MFMA cycles.  Here, per workgroup (8 waves, one per CU,
// persistent over a CONTIGUOUS run of 128-pixel tiles of the input grid):
//   * the weights of all 9 taps (36 KB) are DMA'd once and stay in LDS;
//   * the input lives in a RING of 128 + 2W + 2 (+ 256) grid pixels x 32 channels x 2 planes: a tap is a row offset
//     (kh-PH)*W + (kw-PW) into it, and a new tile only adds the 128 grid pixels behind the previous window (16 KB
//     instead of 9 x 16 KB per tile), fetched two tiles ahead;
//   * waves 0-3 (one per SIMD) do nothing but fragment reads and MFMAs -- 54 back-to-back MFMAs per tile, the next
//     tap's fragments requested before the current tap's MFMAs -- and hand the combined fp32 accumulators to LDS;
//   * waves 4-7 issue the DMA (an LDS-DMA instruction holds its wave ~125 cycles once the queue is full) and run
//     the epilogue of the PREVIOUS tile from the hand-off buffer (scale, bias, ReLU, re-split, 16-byte stores),
//     so neither ever stalls the MFMA stream;
//   * one workgroup barrier per tile.
// Ordering per tile `it` (barrier X(it) at the end of the iteration):
//   compute waves:  taps(it) from ring rows [128 it, 128 it + R16)  ->  hand-off buffer it & 1
//   service waves:  epilogue(it-1) from hand-off (it-1) & 1;  DMA of the rows of tile it+2 (they replace the oldest
//                   128 rows of tile it-1, whose taps ended before X(it-1));  vmcnt(4): the rows of tile it+1 have
//                   landed (loads return in order, and the only younger loads are the four just issued; the
//                   epilogue's stores were issued BEFORE them, so they can only make the wait longer)
// Border handling as the other window kernels (valid: grid pixels without an output are computed and dropped;
// padded: fragments masked per lane and tap).
template <int KHC, int KWC>
__global__ __launch_bounds__(512, 1) void conv_win32_kernel(const ConvArgs p, const int R16, const long long ntiles,
                                                            const int n0) {
    constexpr int BN = 32;
    constexpr int B_PLANE = BN * 64;                      // one plane of one tap's weight tile
    constexpr int B_TAP = 2 * B_PLANE;
    constexpr int ntaps = KHC * KWC;                      // compile-time filter: the tap loop is fully unrolled
    constexpr int HAND = conv_epi::Staging<1>::BYTES;     // hand-off bytes per wave and buffer (>= 4 KB; doubles as staging)
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ring = R16 + 256;                           // rows per plane
    const int win_plane = ring * 64, win_bytes = 2 * win_plane;
    unsigned char* bres = lds;                            // resident weights
    unsigned char* wbuf = lds + ntaps * B_TAP;            // window ring
    unsigned char* epi_area = wbuf + win_bytes;           // chunk descriptors + scale / bias, prepared once
    unsigned char* hand = epi_area + 2048;                // 2 buffers x 4 waves x HAND

    const long long G = (long long)gridDim.x;
    const long long slot = blockIdx.x;
    const long long mgrid = (long long)p.N * p.H * p.W;
    const int minoff = -p.PH * p.W - p.PW;
    const int cl = (lane & 3) ^ ((lane >> 4) & 3);
    const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
    const _Float16* wgt = reinterpret_cast<const _Float16*>(p.w);
    const _Float16* zp = reinterpret_cast<const _Float16*>(g_pipe_zero_page);

    // a workgroup walks a contiguous run of tiles
    const long long per = (ntiles + G - 1) / G;
    const long long t_begin = slot * per, t_end = (t_begin + per < ntiles) ? t_begin + per : ntiles;
    if (t_begin >= ntiles) return;
    const long long g_base = t_begin * 128 + minoff;      // grid pixel of relative row 0
    const long long ntl = t_end - t_begin;

    // resident weights: tap t, plane, 16-row block rb  ->  bres + t * B_TAP + plane * B_PLANE + rb * 1024
    for (int q = wave; q < ntaps * 4; q += 8) {
        const int t = q >> 2, r = q & 3;
        const int plane = r >> 1, rb = r & 1;
        const _Float16* src = wgt + (plane ? p.w_plane : 0) + (long long)(n0 + rb * 16 + (lane >> 2)) * p.Kpad + t * CP_BK + cl * 8;
        unsigned char* dst = bres + t * B_TAP + plane * B_PLANE + rb * 1024;
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);
    }
// DMA of NPC 16-row pieces per plane starting at relative row REL0 (relative to g_base; a multiple of 16);
// physical row = relative row mod ring
#define W32_ROWS(REL0, NPC, Q0, QS)                                                                       \
    {                                                                                                     \
        for (int q = (Q0); q < 2 * (NPC); q += (QS)) {                                                     \
            const int plane = q >= (NPC) ? 1 : 0;                                                          \
            const int rb = q - plane * (NPC);                                                              \
            const long long rel = (REL0) + rb * 16;                                                        \
            const long long g = g_base + rel + (lane >> 2);                                                \
            const bool ok = g >= 0 && g < mgrid;                                                           \
            const _Float16* src = xg + (plane ? p.x_plane : 0) + g * p.Cin + cl * 8;                       \
            src = ok ? src : zp;                                                                           \
            unsigned char* dst = wbuf + plane * win_plane + (int)(rel % ring) * 64;                        \
            __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)dst, 16, 0, 0);                               \
        }                                                                                                  \
    }
    // window of the first tile and the new rows of the second (R16 + 128 rows), by all eight waves
    W32_ROWS(0, (R16 >> 4) + 8, wave, 8)
    {
        conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
        if (tid < BN / 4) {
            sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
            bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
        }
        static_assert(conv_epi::EpiArea<BN>::BYTES <= 2048, "epilogue area");
        conv_epi::prepare<BN>(p, epi_area, n0, sc_pre, bs_pre);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    if (wave >= 4) {
        // ---- service waves: DMA two tiles ahead, epilogue one tile behind ------------------------------------
        const int cw = wave - 4;                           // the compute wave whose tiles this wave finishes
        for (long long it = 0; it <= ntl; ++it) {          // iteration ntl only drains the last epilogue
            if (it >= 1 && !(p.nseg & 0x400)) {
                const unsigned char* hb = hand + ((it - 1) & 1) * 4 * HAND + cw * HAND;
                float16_t am[1][1], ac[1][1];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const conv_epi::float4_t v = *reinterpret_cast<const conv_epi::float4_t*>(hb + (g * 64 + lane) * 16);
#pragma unroll
                    for (int k = 0; k < 4; ++k) { am[0][0][4 * g + k] = v[k]; ac[0][0][4 * g + k] = 0.f; }
                }
                // the hand-off bytes of this wave are consumed (LDS operations of a wave execute in order): reuse them
                // as the staging tile of the stores
                conv_epi::store_tiles_desc<1, 1, true>(p, am, ac, const_cast<unsigned char*>(hb), epi_area,
                                                       (t_begin + it - 1) * 128 + cw * 32);
            }
            if (it + 2 < ntl) { W32_ROWS((long long)R16 + (it + 1) * 128, 8, cw, 4) }
            if (it < ntl) {
                if (it + 2 < ntl) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();              // X(it)
                asm volatile("" ::: "memory");
            }
        }
        return;
    }
    // ---- compute waves ------------------------------------------------------------------------------------------
    const int fswz = ((lane & 31) >> 2) & 3;
    const int fb0 = (lane & 31) * 64 + ((lane >> 5) ^ fswz) * 16;
    const int fb1 = (lane & 31) * 64 + ((2 + (lane >> 5)) ^ fswz) * 16;
    const int lrow0 = wave * 32 + (lane & 31) - minoff;   // window row of this lane's tile row at offset 0
    const unsigned hw = (unsigned)(p.H * p.W);
    for (long long it = 0; it < ntl; ++it) {
        const long long tile = t_begin + it;
        unsigned tapmask = 0xffffffffu;                    // per-lane tap validity (padded convolutions)
        if (p.PH | p.PW) {
            const unsigned g = (unsigned)(tile * 128) + wave * 32 + (lane & 31);      // grid pixels < 2^31 (launcher)
            const unsigned rem = g % hw;
            const int y = (int)(rem / (unsigned)p.W), x = (int)rem - y * p.W;
            tapmask = 0u;
#pragma unroll
            for (int t = 0; t < ntaps; ++t) {
                const int yy = y + t / KWC - p.PH, xx = x + t % KWC - p.PW;
                if (yy >= 0 && yy < p.H && xx >= 0 && xx < p.W) tapmask |= 1u << t;
            }
        }
        float16_t acc_main, acc_corr;
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc_main[j] = 0.f; acc_corr[j] = 0.f; }
        const int wstart = (int)((it * 128) % ring);       // physical row of this tile's window row 0
        half8_t fa_[2][2][2], fb_[2][2][2];
#define W32_READS(TAP, BUF)                                                                               \
        {                                                                                                  \
            const int kh_ = (TAP) / KWC, kw_ = (TAP) % KWC;                                                \
            int wrow = wstart + lrow0 + (kh_ - p.PH) * p.W + (kw_ - p.PW);                                 \
            wrow = wrow >= ring ? wrow - ring : wrow;                                                      \
            const int aswz = (wrow >> 2) & 3;                                                              \
            const unsigned char* ap = wbuf + wrow * 64;                                                    \
            const unsigned char* bb = bres + (TAP) * B_TAP;                                                \
            const unsigned am = (tapmask >> (TAP)) & 1u ? 0xffffffffu : 0u;                                \
            _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                \
                const int ao = ((2 * s + (lane >> 5)) ^ aswz) * 16;                                        \
                u32x4_t ah = *reinterpret_cast<const u32x4_t*>(ap + ao);                                   \
                u32x4_t al = *reinterpret_cast<const u32x4_t*>(ap + win_plane + ao);                       \
                ah &= am; al &= am;                                                                        \
                fa_[BUF][s][0] = __builtin_bit_cast(half8_t, ah);                                          \
                fa_[BUF][s][1] = __builtin_bit_cast(half8_t, al);                                          \
                const unsigned char* bp = bb + (s ? fb1 : fb0);                                            \
                fb_[BUF][s][0] = *reinterpret_cast<const half8_t*>(bp);                                    \
                fb_[BUF][s][1] = *reinterpret_cast<const half8_t*>(bp + B_PLANE);                          \
            }                                                                                              \
        }
#define W32_MFMAS(BUF)                                                                                    \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                    \
            acc_corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[BUF][s][1], fa_[BUF][s][0], acc_corr, 0, 0, 0); \
            acc_main = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[BUF][s][0], fa_[BUF][s][0], acc_main, 0, 0, 0); \
            acc_corr = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[BUF][s][0], fa_[BUF][s][1], acc_corr, 0, 0, 0); \
        }
        if (!(p.nseg & 0x200)) {
        W32_READS(0, 0)
#pragma unroll
        for (int tap = 0; tap < ntaps; ++tap) {
            if (tap + 1 < ntaps) {
                if ((tap + 1) & 1) { W32_READS(tap + 1, 1) } else { W32_READS(tap + 1, 0) }
            }
            if (tap & 1) { W32_MFMAS(1) } else { W32_MFMAS(0) }
        }
        }
        // hand the combined accumulator over: value j of lane l -> float4 slot (j >> 2) * 64 + l
        unsigned char* hb = hand + (it & 1) * 4 * HAND + wave * HAND;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            conv_epi::float4_t v;
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] = acc_main[4 * g + k] + acc_corr[4 * g + k] * (1.0f / 2048.0f);
            *reinterpret_cast<conv_epi::float4_t*>(hb + (g * 64 + lane) * 16) = v;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // X(it)
        asm volatile("" ::: "memory");
    }
}

int launch_win32(const ConvArgs* a, hipStream_t st) {
    if (a->Cin != 32 || a->SH != 1 || a->SW != 1 || a->KH != 3 || a->KW != 3 || a->Kpad != 9 * 32 || a->W < 8 ||
        (long long)a->N * a->H * a->W >= 0x7fffff00LL)
        return TISE_ERR_INVALID_ARG;
    const int R = 128 + 2 * a->W + 2;
    const int R16 = (R + 15) & ~15;
    const size_t lds = 9 * 32 * 128 + (size_t)(R16 + 256) * 128 + 2048 + 8 * (size_t)conv_epi::Staging<1>::BYTES;
    if (lds > 160 * 1024) return TISE_ERR_UNSUPPORTED;
    static size_t attr_lds = 0;
    if (lds > attr_lds) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_win32_kernel<3, 3>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_lds = lds;
    }
    const long long mg = (long long)a->N * a->H * a->W;
    const long long ntiles = (mg + 127) / 128;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        TISE_HIP_CHECK(hipGetDevice(&dev));
        TISE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long grid = ntiles < ncu ? ntiles : ncu;
    for (int n0 = 0; n0 < a->Cout; n0 += 32) {            // 32 couts per launch (the weights of 9 taps x 32 couts stay in LDS)
        hipLaunchKernelGGL((conv_win32_kernel<3, 3>), dim3((unsigned)grid), dim3(512), lds, st, *a, R16, ntiles, n0);
    }
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}


// ------------------------------------------------------------------------------------------------
// Wave-specialised persistent kernel.  What made conv_win32_kernel pay is taken to the general case (Cin % 32 == 0):
// the ablations show DMA-only time ~= MFMA-only time ~= 55-65 % of a launch of the 2-stage kernel and the two
// adding up, because every wave does both -- an LDS-DMA instruction holds its wave ~125 cycles once the queue is
// full, and that is on the critical path of the wave's MFMAs.  Here a persistent workgroup of 8 waves splits:
//   waves 0-3 (one per SIMD): fragment reads + MFMAs of a 128-pixel x BN-cout tile, then the tile's epilogue from
//              their own staging area (conv_epilogue.h, descriptors written by the service waves);
//   waves 4-7: the DMA cursor, two K-steps ahead across tile boundaries (three LDS stages), one counted vmcnt and
//              one barrier per K-step; at a tile boundary they also stage scale / bias / destination descriptors
//              of the tile they are entering (double-buffered: the compute waves are two steps behind).
// Ordering per K-step g:  service: vmcnt(L) [stage g landed: loads return in order, only the L pieces of step g+1 are
// younger] -> barrier X(g) -> DMA of step g+2 into stage (g+2) % 3 = (g-1) % 3 (read in step g-1, which every compute
// wave finished before X(g)).  compute: barrier X(g) -> reads + MFMAs of stage g % 3.
template <int WN, int TMW, int TNW, int CW>
__global__ __launch_bounds__(64 * (CW + 4), 1) void conv_spec_kernel(const ConvArgs p, const int tiles_n, const long long ntiles) {
    // CW compute waves (4: one per SIMD, 128-pixel tiles; 8: two per SIMD, 256-pixel tiles) + 4 service waves
    constexpr int WM = CW / WN;
    constexpr int BM = 32 * TMW * WM;
    static_assert(BM == 32 * CW, "128 pixels per four compute waves");
    constexpr int NJ = BM / 64;                           // pixel rows per service-wave lane
    constexpr int BN = 32 * TNW * WN;
    constexpr int A_PLANE = BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int NBP = BN / 32;                          // weight DMA pieces per service wave and stage
    constexpr int L = 2 * NJ + NBP;
    // epilogue staging: CW = 4: a dedicated area (two tiles wide); CW = 8: the stage the tile's last step consumed
    // (the service waves refill it only after the barrier the compute waves reach after their epilogue)
    constexpr int ETW = (CW == 4 && TNW > 1) ? 2 : 1;
    static_assert(CW == 4 || CW * conv_epi::Staging<1>::BYTES <= STAGE, "staging must fit a stage");
    constexpr int EPI = 2048;                             // bytes of one descriptor / scale / bias area
    static_assert(conv_epi::EpiArea<BN>::BYTES <= EPI, "epilogue area");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    unsigned char* epi_area = lds + 3 * STAGE;            // four areas (tile index & 3): the service waves run up to three tiles ahead
    unsigned char* staging = epi_area + 4 * EPI;          // 4 compute waves x Staging<ETW>::BYTES

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const long long G = (long long)gridDim.x;
    long long slot = blockIdx.x;
    {
        const long long q = G >> 3, r = G & 7, xcd = slot & 7, idx = slot >> 3;
        slot = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int ncb = p.Cin / CP_BK;
    const int nsteps = p.KH * p.KW * ncb;
    const long long my_tiles = slot < ntiles ? (ntiles - slot + G - 1) / G : 0;
    const long long total = my_tiles * nsteps;
    if (total == 0) return;

    if (wave >= CW) {
        // ================================ service waves: DMA cursor ======================================
        const int sw = wave - CW;
        const int stid = tid - 64 * CW;
        const int cl = (lane & 3) ^ ((lane >> 4) & 3);
        const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
        const _Float16* wgt = reinterpret_cast<const _Float16*>(p.w);
        const _Float16* zp = reinterpret_cast<const _Float16*>(g_pipe_zero_page);
        const unsigned ohw = (unsigned)(p.OH * p.OW), M32 = (unsigned)p.M;
        long long is_v = slot;
        int kh = 0, kw = 0, cb = 0, is_stage = 0, tile_par = 0;
        long long issued = 0;
        int ih0[NJ], iw0[NJ];
        const _Float16* img[NJ];
        bool rok[NJ];
        const unsigned char* pa_hi[NJ];
        const unsigned char* pa_lo[NJ];
        long long pa_inc[NJ];
        const _Float16* pb[NBP];
        int pb_off[NBP];
        long long pb_row[NBP];
#pragma unroll
        for (int i = 0; i < NBP; ++i) {
            const int q = sw * NBP + i;                   // 4 * NBP = BN / 8 pieces: 2 planes x BN / 16 row blocks
            const int plane = q >= BN / 16 ? 1 : 0;
            const int rb = q - plane * (BN / 16);
            pb_row[i] = (plane ? p.w_plane : 0) + (long long)(rb * 16 + (lane >> 2)) * p.Kpad + cl * 8;
            pb_off[i] = 2 * A_PLANE + plane * B_PLANE + rb * 1024;
        }
#define CS_TAP()                                                                                          \
        _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                                \
            const int ih = ih0[jj] + kh, iw = iw0[jj] + kw;                                                \
            const bool ok = rok[jj] && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;                         \
            const _Float16* src = img[jj] + ((long long)ih * p.W + iw) * p.Cin;                            \
            pa_hi[jj] = reinterpret_cast<const unsigned char*>(ok ? src : zp);                             \
            pa_lo[jj] = reinterpret_cast<const unsigned char*>(ok ? src + p.x_plane : zp);                 \
            pa_inc[jj] = ok ? CP_BK * 2 : 0;                                                               \
        }
// enter tile is_v: pixel decode (32-bit: M < 2^31, launcher), weight pointers, and the epilogue descriptors of the tile
#define CS_TILE()                                                                                         \
        {                                                                                                  \
            const unsigned tm_ = (unsigned)(is_v / tiles_n);                                               \
            const int tn_ = (int)(is_v - (long long)tm_ * tiles_n);                                        \
            _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                            \
                const unsigned pix = tm_ * BM + (NJ * sw + jj) * 16 + (lane >> 2);                         \
                rok[jj] = pix < M32;                                                                       \
                const unsigned pp = rok[jj] ? pix : 0u;                                                    \
                const unsigned n = pp / ohw;                                                               \
                const unsigned rem = pp - n * ohw;                                                         \
                const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;                  \
                ih0[jj] = (int)oh * p.SH - p.PH;                                                           \
                iw0[jj] = (int)ow * p.SW - p.PW;                                                           \
                img[jj] = xg + (long long)n * p.H * p.W * p.Cin + cl * 8;                                  \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NBP; ++i)                                                \
                pb[i] = wgt + (long long)tn_ * BN * p.Kpad + pb_row[i];                                    \
            {                                                                                              \
                conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};           \
                if (stid < BN / 4) {                                                                       \
                    sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + tn_ * BN + 4 * stid);  \
                    bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + tn_ * BN + 4 * stid);   \
                }                                                                                          \
                conv_epi::prepare<BN>(p, epi_area + tile_par * EPI, tn_ * BN, sc_pre, bs_pre, stid);       \
                tile_par = (tile_par + 1) & 3;                                                             \
            }                                                                                              \
            kh = 0; kw = 0; cb = 0;                                                                        \
            CS_TAP()                                                                                       \
        }
#define CS_ISSUE()                                                                                        \
        {                                                                                                  \
            unsigned char* sb_ = lds + is_stage * STAGE;                                                   \
            _Pragma("unroll") for (int jj = 0; jj < NJ; ++jj) {                                            \
                const unsigned char* sh = pa_hi[jj];                                                       \
                const unsigned char* sl = pa_lo[jj];                                                       \
                unsigned char* da_ = sb_ + (NJ * sw + jj) * 1024;                                          \
                __builtin_amdgcn_global_load_lds(sh, (lds_ptr_t)da_, 16, 0, 0);                            \
                __builtin_amdgcn_global_load_lds(sl, (lds_ptr_t)(da_ + A_PLANE), 16, 0, 0);                \
                pa_hi[jj] = sh + pa_inc[jj];                                                               \
                pa_lo[jj] = sl + pa_inc[jj];                                                               \
            }                                                                                              \
            _Pragma("unroll") for (int i = 0; i < NBP; ++i) {                                              \
                const _Float16* sw_ = pb[i];                                                               \
                unsigned char* dw_ = sb_ + pb_off[i];                                                      \
                __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)dw_, 16, 0, 0);                           \
                pb[i] = sw_ + CP_BK;                                                                       \
            }                                                                                              \
            is_stage = is_stage == 2 ? 0 : is_stage + 1;                                                   \
            ++issued;                                                                                      \
            if (++cb == ncb) {                                                                             \
                cb = 0;                                                                                    \
                if (++kw == p.KW) { kw = 0; ++kh; }                                                        \
                if (kh == p.KH) {                                                                          \
                    is_v += G;                                                                             \
                    if (issued < total) CS_TILE()                                                          \
                } else {                                                                                   \
                    CS_TAP()                                                                               \
                }                                                                                          \
            }                                                                                              \
        }
        CS_TILE()
        CS_ISSUE()
        if (issued < total) CS_ISSUE()
        for (long long g = 0; g < total; ++g) {
            if (issued > g + 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(L) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");            // descriptor writes of CS_TILE
            __builtin_amdgcn_s_barrier();                                 // X(g)
            asm volatile("" ::: "memory");
            if (issued < total) CS_ISSUE()
            if (CW == 8 && (g + 1) % nsteps == 0) {                       // E: see the compute waves' tile end
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        }
        return;
    }
    // ==================================== compute waves ==============================================
    const int wm = wave / WN, wn = wave - wm * WN;
    float16_t acc_main[TMW][TNW], acc_corr[TMW][TNW];
    CP_ZERO()
    const int fswz = ((lane & 31) >> 2) & 3;
    const int fo0 = (lane & 31) * 64 + (((lane >> 5)) ^ fswz) * 16;
    const int fo1 = (lane & 31) * 64 + ((2 + (lane >> 5)) ^ fswz) * 16;
    const int fa_off = wm * TMW * 32 * 64;
    const int fb_off = 2 * A_PLANE + wn * TNW * 32 * 64;
    constexpr int NREADS = 4 * (TMW + TNW), NGROUPS = 2 * TMW * TNW;
    constexpr int NR2 = (NREADS - 4) / 2;
    long long c_v = slot;
    int c_step = 0, c_stage = 0, c_par = 0;
    for (long long g = 0; g < total; ++g) {
        __builtin_amdgcn_s_barrier();                                     // X(g)
        asm volatile("" ::: "memory");
        const unsigned char* sb = lds + c_stage * STAGE;
        CP_COMPUTE(sb)
        c_stage = c_stage == 2 ? 0 : c_stage + 1;
        if (++c_step == nsteps) {
            if (CW == 8) {                                                // E: every compute wave has read its last fragments
                __builtin_amdgcn_s_barrier();                             // before the consumed stage becomes staging
                asm volatile("" ::: "memory");
            }
            const long long tm_ = c_v / tiles_n;
            const int tn_ = (int)(c_v - tm_ * tiles_n);
            // a wave owns TMW row tiles: one call per row tile (store_tiles_desc handles one 32-row strip)
#pragma unroll
            for (int i = 0; i < TMW; ++i) {
                float16_t am[1][TNW], ac[1][TNW];
#pragma unroll
                for (int t = 0; t < TNW; ++t) { am[0][t] = acc_main[i][t]; ac[0][t] = acc_corr[i][t]; }
                unsigned char* stg = CW == 4 ? staging + wave * conv_epi::Staging<ETW>::BYTES
                                             : lds + (c_stage == 0 ? 2 : c_stage - 1) * STAGE + wave * conv_epi::Staging<ETW>::BYTES;
                conv_epi::store_tiles_desc<TNW, ETW, false, BN>(p, am, ac, stg,
                                                                epi_area + c_par * EPI, tm_ * BM + (wm * TMW + i) * 32, wn * TNW * 4);
            }
            (void)tn_;
            CP_ZERO()
            c_step = 0;
            c_v += G;
            c_par = (c_par + 1) & 3;
        }
    }
}

template <int WN, int TMW, int TNW, int CW>
int launch_spec(const ConvArgs* a, hipStream_t st) {
    constexpr int BN = 32 * TNW * WN;
    constexpr int BM = 32 * CW;
    constexpr int STAGE = 2 * BM * 64 + 2 * BN * 64;
    constexpr int ETW = (CW == 4 && TNW > 1) ? 2 : 1;
    constexpr int LDS = 3 * STAGE + 4 * 2048 + (CW == 4 ? 4 * conv_epi::Staging<ETW>::BYTES : 0);
    static_assert(LDS <= 160 * 1024, "LDS budget");
    if (a->Cin % 32 != 0 || a->Kpad != a->KH * a->KW * a->Cin || a->M >= 0x7fffff00LL) return TISE_ERR_INVALID_ARG;
    static bool attr_set = false;
    if (!attr_set) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_spec_kernel<WN, TMW, TNW, CW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        attr_set = true;
    }
    const int tiles_n = (a->Cout + BN - 1) / BN;
    const long long ntiles = ((a->M + BM - 1) / BM) * tiles_n;
    static int ncu = 0;
    if (ncu == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        TISE_HIP_CHECK(hipGetDevice(&dev));
        TISE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
        ncu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    }
    const long long grid = ntiles < ncu ? ntiles : ncu;
    hipLaunchKernelGGL((conv_spec_kernel<WN, TMW, TNW, CW>), dim3((unsigned)grid), dim3(64 * (CW + 4)), LDS, st, *a, tiles_n, ntiles);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // namespace

// cfg: tile width / wave layout / schedule.  Ping-pong schedule: 0: 128 couts (4x2 waves of 64x64)  1: 96 (8x1 of
// 32x96)  2: 64 (8x1 of 32x64)  4: 64 (4x2 of 64x32)  5: 32 (8x1 of 32x32).  Lockstep schedule (all waves in
// the same phase): 8, 9, 10 = the layouts of 0, 1, 2;  3: 160 (8x1 of 32x160)  6: 128 (8x1 of 32x128).
// Resident-weights sliding-window kernel for Cin = 32, 3x3, stride 1: 33 (32 couts per launch).
// Wave-specialised persistent kernel (Cin % 32 == 0), 128-pixel tiles: 40 (128 couts, 2x2 compute waves), 41 (96), 42 (64),
// 43 (128, 4x1), 44 (160).
// Window kernel (stride 1, more than one tap): 11, 12, 13, 14 = 128, 96, 64, 32 couts (ping-pong);
// 7, 15 = 128, 96 couts lockstep.
int tise_conv_pipe_launch(const tise_conv_args* a, int cfg, void* stream) {
    if (a->KH * a->KW * ((a->Cin + 31) / 32) < 1) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2, true>(a, st);
        case 1: return launch_cfg<1, 1, 3, true>(a, st);
        case 2: return launch_cfg<1, 1, 2, true>(a, st);
        case 4: return launch_cfg<2, 2, 1, true>(a, st);
        case 5: return launch_cfg<1, 1, 1, true>(a, st);
        case 8: return launch_cfg<2, 2, 2, false>(a, st);
        case 9: return launch_cfg<1, 1, 3, false>(a, st);
        case 10: return launch_cfg<1, 1, 2, false>(a, st);
        case 3: return launch_cfg<1, 1, 5, false>(a, st);
        case 33: return launch_win32(a, st);
        case 40: return launch_spec<2, 2, 2, 4>(a, st);       // 128 couts, compute waves 2 x 2 of 64 x 64
        case 41: return launch_spec<1, 1, 3, 4>(a, st);       // 96 couts, compute waves 4 x 1 of 32 x 96
        case 42: return launch_spec<1, 1, 2, 4>(a, st);       // 64
        case 43: return launch_spec<1, 1, 4, 4>(a, st);       // 128, 4 x 1 of 32 x 128
        case 44: return launch_spec<1, 1, 5, 4>(a, st);       // 160
        case 45: return launch_spec<2, 2, 2, 8>(a, st);       // 256 pixels x 128 couts: 8 compute waves 4 x 2 of 64 x 64
        case 46: return launch_spec<1, 1, 3, 8>(a, st);       // 256 x 96: 8 x 1 of 32 x 96
        case 47: return launch_spec<1, 1, 4, 8>(a, st);       // 256 x 128: 8 x 1 of 32 x 128
        case 7: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<2, 2, 2, false>(a, st);
        case 15: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<1, 1, 3, false>(a, st);
        case 11: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<2, 2, 2, true>(a, st);
        case 12: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<1, 1, 3, true>(a, st);
        case 13: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<1, 1, 2, true>(a, st);
        case 14: if (a->SH != 1 || a->SW != 1) return TISE_ERR_INVALID_ARG; return launch_win<1, 1, 1, true>(a, st);
        case 6: return launch_cfg<1, 1, 4, false>(a, st);
        default: return TISE_ERR_INVALID_ARG;
    }
}
