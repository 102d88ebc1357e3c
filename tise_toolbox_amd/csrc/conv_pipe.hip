// Persistent, wave-specialised forms of the split-precision convolution (conv_split.hip explains the arithmetic:
// v ~= hi + lo * 2^-11, three fp16 MFMAs per product, fp32 accumulate; same operand formats, same results up
// to fp32 summation order).  Two kernels live here:
//
//   conv_win32_kernel   (configuration 33)  resident-weights sliding-window kernel for Cin = 32, 3x3, stride 1:
//                       the default for Conv2d_2a (1.34 -> 1.05 ms at batch 500).
//   conv_spec_kernel    (configurations 45-47) 256-pixel tiles, 8 compute + 4 service waves: opt-in per-layer
//                       table TISE_CONV_AUTO=1 (5-13 % faster in isolation on the deep 1x1 layers, bit-identical;
//                       no gain in the bench).
//
// Round 1 also carried a 3-stage persistent kernel (ping-pong and lockstep schedules, configurations 0-10), its
// window-resident form (7, 11-15) and 128-pixel wave-specialised configurations (40-44).  All of them measured
// within +-3 % of the default `fast` kernel (profiles/r01g_conv_pipe_probe.txt, r01i_conv_spec_probe.txt) and were
// removed in round 2 together with their tests; the measurements stay in profiles/ and DESIGN.md.
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

// accumulator reset and one K-step of fragment reads + MFMAs; the using kernel defines acc_main / acc_corr
// [TMW][TNW], fo0 / fo1 (swizzled fragment offsets), fa_off / fb_off, A_PLANE / B_PLANE, NGROUPS and NR2.
#define CP_ZERO()                                                                                         \
    _Pragma("unroll") for (int i = 0; i < TMW; ++i)                                                        \
        _Pragma("unroll") for (int t = 0; t < TNW; ++t)                                                    \
            _Pragma("unroll") for (int j = 0; j < 16; ++j) { acc_main[i][t][j] = 0.f; acc_corr[i][t][j] = 0.f; }
// all fragment reads written first, the MFMAs after them, and a sched_group_barrier sequence that makes the
// scheduler emit "2 reads, 3 MFMAs" alternately after the first 4 reads
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
    static std::atomic<unsigned long long> attr_set{0};
    if (tise_first_use_on_this_device(attr_set)) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_spec_kernel<WN, TMW, TNW, CW>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
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

// cfg 33: resident-weights sliding-window kernel (Cin = 32, 3x3, stride 1; 32 couts per launch).
// cfg 45 / 46 / 47: wave-specialised persistent kernel (Cin % 32 == 0), 256-pixel tiles x 128 / 96 / 128 couts
// (compute waves 4 x 2 of 64 x 64, 8 x 1 of 32 x 96, 8 x 1 of 32 x 128).
int tise_conv_pipe_launch(const tise_conv_args* a, int cfg, void* stream) {
    if (a->KH * a->KW * ((a->Cin + 31) / 32) < 1) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    switch (cfg) {
        case 33: return launch_win32(a, st);
        case 45: return launch_spec<2, 2, 2, 8>(a, st);
        case 46: return launch_spec<1, 1, 3, 8>(a, st);
        case 47: return launch_spec<1, 1, 4, 8>(a, st);
        default: return TISE_ERR_INVALID_ARG;
    }
}

TISE_DEFINE_SPLIT_FLAG_READER(tise_internal_split_flag_conv_pipe)
