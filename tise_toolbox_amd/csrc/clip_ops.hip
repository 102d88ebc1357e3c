// Hand-written kernels of the CLIP ViT-B/32 towers (text_relevance/RP_coco.py:56-80 and
// positional_alignment/PA.py:33-43 call the third-party `clip` model once per item; here the towers run batched).
// fp16 tensors, fp32 accumulation / statistics, as the fp16 model `clip.load` serves on a GPU computes.
//
//   gemm_f16_kernel      out = act(A W^T + bias) + residual      nn.Linear / the patch-embedding conv / the projections
//   gemm_f16_big_kernel  128 x 128 x 64 tiles, 4 waves (64 x 64 per wave), two workgroups per CU; for launches of many
//                        tiles 256 x 256 x 64, 8 waves (128 x 64 per wave), one workgroup per CU.  Both:
//                        v_mfma_f32_32x32x16_f16, operands global -> LDS by global_load_lds_dwordx4 in 128-BYTE rows
//                        (a K-step of 64 halves is one line), two stages, XOR-swizzled LDS rows (conflict-free
//                        ds_read_b128), epilogue staged per wave so that rows leave as full lines.
//   layernorm_f16_kernel one wave per row, fp32 mean / variance (CLIP's LayerNorm computes in fp32), eps inside sqrt.
//   attention_f16_kernel one WAVE per (sequence, head): S <= 96 tokens, head dim 64: K Q^T and V^T P^T on the matrix
//                        cores with operands loaded straight into the MFMA layout, fp32 softmax in registers.
//   vit_tokens_f16_kernel / text_tokens_f16_kernel / patchify_f16_kernel / gather_rows_f16_kernel: token assembly.
#include <hip/hip_fp16.h>
#include <cstdlib>
#include "common.h"

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __attribute__((aligned(128))) unsigned char g_clip_zero_page[128];

#define GM_BK 64

struct GemmArgs {
    const _Float16* a; long long lda;
    const _Float16* w; long long ldw;
    const _Float16* bias;                            // [N] or null
    const _Float16* res; long long ldr;              // [M][ldr] or null
    _Float16* out; long long ldo;
    int M, N, K, act;                                // act: 0 none, 1 QuickGELU (x * sigmoid(1.702 x))
};

// 128 x 128 x 64 tiles: four waves (2 x 2 of 64 x 64), TWO LDS stages of 32 KB, two workgroups per CU -- the structure of
// the convolution kernel (conv_split.hip).  The first version used 256 x 128 tiles with eight waves and three stages
// (144 KB of LDS: one workgroup per CU), so the DMA latency in front of a tile's first K-step and its epilogue were
// fully exposed, and the towers' GEMMs are short: K = 512 / 768 is 8 / 12 K-steps.  With two workgroups per CU the
// second one computes meanwhile: 5-15 % faster on every tower shape (tools/clip_gemm_probe.py) although a CU now
// moves 64 KB instead of 48 KB of operands per 128 MFMAs.
__global__ __launch_bounds__(256, 2) void gemm_f16_kernel(const GemmArgs p) {
    constexpr int BM = 128, BN = 128;
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;         // 32 KB
    __shared__ __attribute__((aligned(1024))) unsigned char lds[2 * STAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const unsigned tiles_n = (unsigned)(p.N + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (int)(bid / tiles_n) * BM;
    const int n0 = (int)(bid % tiles_n) * BN;
    const unsigned char* zp = g_clip_zero_page;
    // DMA pieces (8 rows x 128 B): this wave fetches rows 32*wave .. +31 of both operands
    const unsigned char* pa[4];
    const unsigned char* pw[4];
    int ia[4], iw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = (4 * wave + g) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const bool oka = m0 + r < p.M, okw = n0 + r < p.N;
        pa[g] = oka ? reinterpret_cast<const unsigned char*>(p.a + (long long)(m0 + r) * p.lda + c * 8) : zp;
        ia[g] = oka ? 128 : 0;
        pw[g] = okw ? reinterpret_cast<const unsigned char*>(p.w + (long long)(n0 + r) * p.ldw + c * 8) : zp;
        iw[g] = okw ? 128 : 0;
    }
#define GS_ISSUE(SOFF)                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = pa[g];                                                               \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + (4 * wave + g) * 1024), 16, 0, 0);            \
            pa[g] = s_ + ia[g];                                                                            \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = pw[g];                                                               \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + A_BYTES + (4 * wave + g) * 1024), 16, 0, 0);  \
            pw[g] = s_ + iw[g];                                                                            \
        }                                                                                                  \
    }
    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fsw = (frow >> 1) & 7;
    int fo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fo[s] = frow * 128 + (((2 * s + (lane >> 5)) ^ fsw) * 16);
    const unsigned char* fa = lds + (wm * 64) * 128;
    const unsigned char* fb = lds + A_BYTES + (wn * 64) * 128;
// the first K-slice's fragments are requested before the step's DMA is issued (LDS latency under the issue phase)
#define GS_HEAD(SOFF)                                                                                     \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                    \
            h_a[i] = *reinterpret_cast<const half8_t*>(fa + (SOFF) + i * 32 * 128 + fo[0]);                \
            h_b[i] = *reinterpret_cast<const half8_t*>(fb + (SOFF) + i * 32 * 128 + fo[0]);                \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define GS_COMPUTE(SOFF)                                                                                  \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        half8_t a_[4][2], b_[4][2];                                                                        \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i) {                                                \
                if (s == 0) { a_[0][i] = h_a[i]; b_[0][i] = h_b[i]; }                                      \
                else {                                                                                     \
                    a_[s][i] = *reinterpret_cast<const half8_t*>(fa + (SOFF) + i * 32 * 128 + fo[s]);      \
                    b_[s][i] = *reinterpret_cast<const half8_t*>(fb + (SOFF) + i * 32 * 128 + fo[s]);      \
                }                                                                                          \
            }                                                                                              \
        _Pragma("unroll") for (int s = 0; s < 4; ++s)                                                      \
            _Pragma("unroll") for (int i = 0; i < 2; ++i)                                                  \
                _Pragma("unroll") for (int j = 0; j < 2; ++j)                                              \
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_[s][j], a_[s][i], acc[i][j], 0, 0, 0); \
        /* slice s+1's four reads go out under slice s's four MFMAs */                                     \
        _Pragma("unroll") for (int s = 0; s < 3; ++s) {                                                    \
            __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                             \
        }                                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                                                 \
    }
    half8_t h_a[2], h_b[2];
    const int nsteps = p.K / GM_BK;
    GS_ISSUE(0)
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GS_HEAD(0)
        GS_ISSUE(STAGE)
        GS_COMPUTE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GS_HEAD(STAGE)
        if (step + 2 < nsteps) GS_ISSUE(0)
        GS_COMPUTE(STAGE)
    }
    if (step < nsteps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GS_HEAD(0)
        GS_COMPUTE(0)
    }
    __syncthreads();
    constexpr int PITCH = 144;
    unsigned char* st = lds + wave * (32 * PITCH);
    const int ncol0 = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = j * 32 + 8 * g + 4 * (lane >> 5);
                const int n = ncol0 + cl;
                half4_t h, bq = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                if (p.bias && n < p.N) bq = *reinterpret_cast<const half4_t*>(p.bias + n);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[i][j][4 * g + k] + (float)bq[k];
                    if (p.act == 1) v = v / (1.0f + __expf(-1.702f * v));
                    h[k] = (_Float16)v;
                }
                *reinterpret_cast<half4_t*>(st + (lane & 31) * PITCH + cl * 2) = h;
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + wm * 64 + i * 32 + r, n = ncol0 + ch * 8;
            half8_t v = *reinterpret_cast<const half8_t*>(st + r * PITCH + ch * 16);
            if (m < p.M && n < p.N) {
                if (p.res) {
                    const half8_t rr = *reinterpret_cast<const half8_t*>(p.res + (long long)m * p.ldr + n);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (_Float16)((float)v[k] + (float)rr[k]);
                }
                *reinterpret_cast<half8_t*>(p.out + (long long)m * p.ldo + n) = v;
            }
        }
    }
}

// 256 x 256 x 64 tiles for the large GEMMs (at least four tiles per CU): eight waves (2 x 4, 128 x 64 per wave), two
// stages of 64 KB, one workgroup per CU.  A CU moves 64 KB of operands per 256 MFMAs here -- 32 B/clk against the
// 64 B/clk of the 128 x 128 kernel and the ~38 B/clk the LDS-DMA path delivers -- and a wave reads 6 fragments per 8
// MFMAs instead of 4 per 4.  The price is one workgroup per CU: the DMA latency in front of a tile's first K-step and
// its epilogue are exposed, so the launcher uses it only where a tile has enough K-steps and the launch enough tiles.
__global__ __launch_bounds__(512, 1) void gemm_f16_big_kernel(const GemmArgs p) {
    constexpr int BM = 256, BN = 256;
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;         // 64 KB
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                   // 2 x 4 waves of 128 x 64
    const unsigned tiles_n = (unsigned)(p.N + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (int)(bid / tiles_n) * BM;
    const int n0 = (int)(bid % tiles_n) * BN;
    const unsigned char* zp = g_clip_zero_page;
    const unsigned char* pa[4];
    const unsigned char* pw[4];
    int ia[4], iw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = (4 * wave + g) * 8 + (lane >> 3);        // 8 waves x 4 pieces x 8 rows = 256 rows of each operand
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        const bool oka = m0 + r < p.M, okw = n0 + r < p.N;
        pa[g] = oka ? reinterpret_cast<const unsigned char*>(p.a + (long long)(m0 + r) * p.lda + c * 8) : zp;
        ia[g] = oka ? 128 : 0;
        pw[g] = okw ? reinterpret_cast<const unsigned char*>(p.w + (long long)(n0 + r) * p.ldw + c * 8) : zp;
        iw[g] = okw ? 128 : 0;
    }
#define GB_ISSUE(SOFF)                                                                                    \
    {                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = pa[g];                                                               \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + (4 * wave + g) * 1024), 16, 0, 0);            \
            pa[g] = s_ + ia[g];                                                                            \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = pw[g];                                                               \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + A_BYTES + (4 * wave + g) * 1024), 16, 0, 0);  \
            pw[g] = s_ + iw[g];                                                                            \
        }                                                                                                  \
    }
    float16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    const int frow = lane & 31, fsw = (frow >> 1) & 7;
    int fo[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) fo[s] = frow * 128 + (((2 * s + (lane >> 5)) ^ fsw) * 16);
    const unsigned char* fa = lds + (wm * 128) * 128;
    const unsigned char* fb = lds + A_BYTES + (wn * 64) * 128;
// reads of slice s+1 are written before the MFMAs of slice s (six reads per eight MFMAs)
#define GB_READS(S, SOFF, BUF)                                                                            \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                      \
            a_[BUF][i] = *reinterpret_cast<const half8_t*>(fa + (SOFF) + i * 32 * 128 + fo[S]);            \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
            b_[BUF][j] = *reinterpret_cast<const half8_t*>(fb + (SOFF) + j * 32 * 128 + fo[S]);            \
    }
#define GB_MFMAS(BUF)                                                                                     \
    _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                      \
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_[BUF][j], a_[BUF][i], acc[i][j], 0, 0, 0);
#define GB_COMPUTE(SOFF)                                                                                  \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        GB_READS(1, SOFF, 1) GB_MFMAS(0) __builtin_amdgcn_sched_barrier(0);                                \
        GB_READS(2, SOFF, 0) GB_MFMAS(1) __builtin_amdgcn_sched_barrier(0);                                \
        GB_READS(3, SOFF, 1) GB_MFMAS(0) __builtin_amdgcn_sched_barrier(0);                                \
        GB_MFMAS(1) __builtin_amdgcn_sched_barrier(0);                                                     \
    }
    half8_t a_[2][4], b_[2][2];
    const int nsteps = p.K / GM_BK;
    GB_ISSUE(0)
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GB_READS(0, 0, 0)
        __builtin_amdgcn_sched_barrier(0);
        GB_ISSUE(STAGE)
        GB_COMPUTE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GB_READS(0, STAGE, 0)
        __builtin_amdgcn_sched_barrier(0);
        if (step + 2 < nsteps) GB_ISSUE(0)
        GB_COMPUTE(STAGE)
    }
    if (step < nsteps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        GB_READS(0, 0, 0)
        GB_COMPUTE(0)
    }
    __syncthreads();
    constexpr int PITCH = 144;
    unsigned char* st = lds + wave * (32 * PITCH);
    const int ncol0 = n0 + wn * 64;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int cl = j * 32 + 8 * g + 4 * (lane >> 5);
                const int n = ncol0 + cl;
                half4_t h, bq = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
                if (p.bias && n < p.N) bq = *reinterpret_cast<const half4_t*>(p.bias + n);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[i][j][4 * g + k] + (float)bq[k];
                    if (p.act == 1) v = v / (1.0f + __expf(-1.702f * v));
                    h[k] = (_Float16)v;
                }
                *reinterpret_cast<half4_t*>(st + (lane & 31) * PITCH + cl * 2) = h;
            }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + wm * 128 + i * 32 + r, n = ncol0 + ch * 8;
            half8_t v = *reinterpret_cast<const half8_t*>(st + r * PITCH + ch * 16);
            if (m < p.M && n < p.N) {
                if (p.res) {
                    const half8_t rr = *reinterpret_cast<const half8_t*>(p.res + (long long)m * p.ldr + n);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (_Float16)((float)v[k] + (float)rr[k]);
                }
                *reinterpret_cast<half8_t*>(p.out + (long long)m * p.ldo + n) = v;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: both GEMM kernels on v_mfma_f32_16x16x32_f16 (gemm16_f16_kernel<BIG>).  Same tiles, LDS images, DMA and
// epilogue staging as the two kernels above; inside a wave a fragment is 16 rows x 32 K (lane -> row lane & 15, 16-byte
// chunk lane >> 4 of the K-step's first or second 64 bytes) and the accumulator four-register blocks of 16 x 16.  The
// convolution kernels made the same move in round 3 (same flop, same LDS bytes, but the K = 32 shape reads and
// writes half the accumulator registers per flop and the chip holds 1.79 instead of 1.52 GHz under its power limit).
// The chunk swizzle is the convolution kernels' tise_lds_swz (conflict-free for 16-row fragments).
// BIG = false: 128 x 128 x 64 tiles, 4 waves of 64 x 64, two workgroups per CU;  BIG = true: 256 x 256 x 64, 8 waves of
// 128 x 64, one per CU.  TISE_GEMM_SHAPE=32 selects the 32x32x16 kernels above (A/B of tools/clip_gemm_probe.py).
typedef float float4_t __attribute__((ext_vector_type(4)));

template <bool BIG>
__global__ __launch_bounds__(BIG ? 512 : 256, BIG ? 1 : 2) void gemm16_f16_kernel(const GemmArgs p) {
    constexpr int BM = BIG ? 256 : 128, BN = BM;
    constexpr int NW = BIG ? 8 : 4;
    constexpr int WM = BIG ? 128 : 64;                          // rows of A per wave; 64 columns (rows of W) per wave
    constexpr int MI = WM / 16, NI = 4;                         // 16-row fragments per wave along m and n
    constexpr int A_BYTES = BM * 128, STAGE = A_BYTES + BN * 128;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = BIG ? (wave >> 2) : (wave >> 1), wn = BIG ? (wave & 3) : (wave & 1);
    const unsigned tiles_n = (unsigned)(p.N + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int m0 = (int)(bid / tiles_n) * BM;
    const int n0 = (int)(bid % tiles_n) * BN;
    // DMA pieces (8 rows x 128 B): this wave fetches rows 32 wave .. +31 of both operands.  Source = one scalar base per
    // operand (advanced by 128 bytes per K-step on the scalar unit) + a constant 32-bit byte offset per lane and piece (the
    // `saddr` form of the instruction: no per-step vector arithmetic, 8 address registers instead of 24).  Rows beyond M
    // (N) are CLAMPED to the last valid row instead of redirected to a zero page: a row of the product depends on its own
    // operand row only, and rows / columns beyond M / N are never stored.
    const unsigned char* abase = reinterpret_cast<const unsigned char*>(p.a + (long long)m0 * p.lda);
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w + (long long)n0 * p.ldw);
    unsigned offa[4], offw[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int r = (4 * wave + g) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ tise_lds_swz(r);
        const int ra = m0 + r < p.M ? r : p.M - 1 - m0, rw = n0 + r < p.N ? r : p.N - 1 - n0;
        offa[g] = (unsigned)ra * (unsigned)p.lda * 2u + c * 16;
        offw[g] = (unsigned)rw * (unsigned)p.ldw * 2u + c * 16;
    }
#define G16_ISSUE(SOFF)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = abase + offa[g];                                                     \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + (4 * wave + g) * 1024), 16, 0, 0);            \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 4; ++g) {                                                    \
            const unsigned char* s_ = wbase + offw[g];                                                     \
            __builtin_amdgcn_global_load_lds(s_, (lds_ptr_t)(lds + (SOFF) + A_BYTES + (4 * wave + g) * 1024), 16, 0, 0);  \
        }                                                                                                  \
        abase += 128; wbase += 128;                                                                        \
    }
    float4_t acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    // fragment: row (lane & 15) of a 16-row block, chunk (lane >> 4) of K-slice 0 (slice 1: chunk + 4 = address ^ 64)
    const int f16o = (lane & 15) * 128 + (((lane >> 4) ^ tise_lds_swz(lane & 15)) << 4);
    const unsigned char* fa = lds + (wm * WM) * 128;
    const unsigned char* fb = lds + A_BYTES + (wn * 64) * 128;
    half8_t a_[MI], b_[2][NI];
// K-slice 0 of the stage: all fragments
#define G16_READS(SOFF)                                                                                   \
    {                                                                                                     \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                     \
            a_[i] = *reinterpret_cast<const half8_t*>(fa + (SOFF) + i * 2048 + f16o);                      \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                     \
            b_[0][j] = *reinterpret_cast<const half8_t*>(fb + (SOFF) + j * 2048 + f16o);                   \
    }
// slice 0's MFMAs row block by row block; a row block's slice-1 fragment is requested into the registers its slice-0
// fragment has just left (the LDS latency passes under the following blocks' MFMAs; 64 + 32 fragment registers live
// instead of 96 + 48: the 256 x 256 instance would spill otherwise), then slice 1's MFMAs
#define G16_COMPUTE(SOFF)                                                                                 \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                     \
            b_[1][j] = *reinterpret_cast<const half8_t*>(fb + (SOFF) + j * 2048 + (f16o ^ 64));            \
        _Pragma("unroll") for (int i = 0; i < MI; ++i) {                                                   \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_[0][j], a_[i], acc[i][j], 0, 0, 0);   \
            a_[i] = *reinterpret_cast<const half8_t*>(fa + (SOFF) + i * 2048 + (f16o ^ 64));               \
            __builtin_amdgcn_sched_barrier(0);                                                             \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < MI; ++i)                                                     \
            _Pragma("unroll") for (int j = 0; j < NI; ++j)                                                 \
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_[1][j], a_[i], acc[i][j], 0, 0, 0);   \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
    const int nsteps = p.K / GM_BK;
    G16_ISSUE(0)
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        G16_READS(0)
        __builtin_amdgcn_sched_barrier(0);
        G16_ISSUE(STAGE)
        G16_COMPUTE(0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        G16_READS(STAGE)
        __builtin_amdgcn_sched_barrier(0);
        if (step + 2 < nsteps) G16_ISSUE(0)
        G16_COMPUTE(STAGE)
    }
    if (step < nsteps) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        G16_READS(0)
        G16_COMPUTE(0)
    }
    __syncthreads();
#undef G16_ISSUE
#undef G16_READS
#undef G16_COMPUTE
    // epilogue: 32 rows x 64 columns at a time through the wave's staging rows (as the kernels above); the accumulator
    // block (i, j) holds row m = 16 i + (lane & 15) and the four columns n = 16 j + 4 (lane >> 4) + k of this lane
    constexpr int PITCH = 144;
    unsigned char* st = lds + wave * (32 * PITCH);
    const int ncol0 = n0 + wn * 64;
#pragma unroll
    for (int i2 = 0; i2 < MI / 2; ++i2) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int cl = 16 * j + 4 * (lane >> 4);
            const int n = ncol0 + cl;
            half4_t bq = {(_Float16)0.f, (_Float16)0.f, (_Float16)0.f, (_Float16)0.f};
            if (p.bias && n < p.N) bq = *reinterpret_cast<const half4_t*>(p.bias + n);
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2) {
                half4_t h;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = acc[2 * i2 + h2][j][k] + (float)bq[k];
                    if (p.act == 1) v = v / (1.0f + __expf(-1.702f * v));
                    h[k] = (_Float16)v;
                }
                *reinterpret_cast<half4_t*>(st + (16 * h2 + (lane & 15)) * PITCH + cl * 2) = h;
            }
        }
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int r = pass * 8 + (lane >> 3), ch = lane & 7;
            const int m = m0 + wm * WM + i2 * 32 + r, n = ncol0 + ch * 8;
            half8_t v = *reinterpret_cast<const half8_t*>(st + r * PITCH + ch * 16);
            if (m < p.M && n < p.N) {
                if (p.res) {
                    const half8_t rr = *reinterpret_cast<const half8_t*>(p.res + (long long)m * p.ldr + n);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (_Float16)((float)v[k] + (float)rr[k]);
                }
                *reinterpret_cast<half8_t*>(p.out + (long long)m * p.ldo + n) = v;
            }
        }
    }
}

// one wave per row: y = (x - mean) / sqrt(var + eps) * gamma + beta, statistics in fp32 (two passes over registers).
// 16-byte loads and stores: lane l holds columns 8l .. 8l+7 and 512 + 8l .. (C <= 1024, C % 8 == 0) -- the first
// version moved 2 bytes per lane and instruction and ran at 2 TB/s.
__global__ __launch_bounds__(256) void layernorm_f16_kernel(const _Float16* __restrict__ x, long long ldx,
                                                            const _Float16* __restrict__ gamma, const _Float16* __restrict__ beta,
                                                            _Float16* __restrict__ out, long long ldo, int rows, int C, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const _Float16* xr = x + (long long)row * ldx;
    float v[2][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = 512 * k + 8 * lane;
        half8_t h = {};
        if (c < C) h = *reinterpret_cast<const half8_t*>(xr + c);
#pragma unroll
        for (int e = 0; e < 8; ++e) { v[k][e] = (float)h[e]; s += v[k][e]; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        if (512 * k + 8 * lane < C) {
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float d = v[k][e] - mean; q += d * d; }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) q += __shfl_xor(q, off, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
    _Float16* o = out + (long long)row * ldo;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int c = 512 * k + 8 * lane;
        if (c < C) {
            const half8_t g = *reinterpret_cast<const half8_t*>(gamma + c);
            const half8_t bb = *reinterpret_cast<const half8_t*>(beta + c);
            half8_t r;
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = (_Float16)((v[k][e] - mean) * rstd * (float)g[e] + (float)bb[e]);
            *reinterpret_cast<half8_t*>(o + c) = r;
        }
    }
}

// qkv: [B*S][3*E] (q | k | v, head h = columns h*64 .. h*64+63 of each), out: [B*S][E].  ONE WAVE per (sequence, head),
// everything on the matrix cores:
//   S^T[j][i] = sum_d K[j][d] Q[i][d]    A operand = K rows, B operand = Q rows: both are 8 consecutive d per lane, i.e.
//                                         16-byte loads straight from global memory into the MFMA operand layout;
//   the accumulator of a lane then holds ONE query i = lane & 31 and keys j = 32 jt + 8 g + 4 (lane >> 5) + k' along
//   its registers: the softmax over the keys is an in-lane reduction plus one exchange with lane ^ 32;
//   O^T[d][i] = sum_j V^T[d][j] P^T[j][i]  B operand = the probabilities exactly as they sit in the registers (fp16);
//                                         A operand = V^T rows from an LDS copy of V transposed on the way in.  The
//   K index of that product is permuted (j runs 4h + k', 8 + 4h + k' inside a 16-slice) consistently on both operands.
// NT = key / query tiles of 32 (2: S <= 64, the image tower's 50; 3: S <= 96, the text tower's 77).
template <int NT>
__global__ __launch_bounds__(256) void attention_f16_kernel(const _Float16* __restrict__ qkv, int S, int H, int BH, int causal,
                                                            _Float16* __restrict__ out) {
    constexpr int SP = 32 * NT, VP = SP + 8;                   // padded key count; V^T pitch in halves (16-byte aligned rows)
    __shared__ __attribute__((aligned(16))) _Float16 vt_all[4][64 * VP];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bh = blockIdx.x * 4 + wave;
    if (bh >= BH) return;
    const int b = bh / H, h = bh - b * H;
    const int E = H * 64;
    const long long ld = 3LL * E;
    const _Float16* qb = qkv + (long long)b * S * ld + h * 64;
    const _Float16* kb = qb + E;
    const _Float16* vb = qb + 2 * E;
    _Float16* vt = vt_all[wave];
    const int r = lane & 31, hh = lane >> 5;
    // ---- V^T into LDS: lane (row j = 8 t + (lane >> 3), 8 d's = lane & 7) ------------------------------------
    for (int j0 = 0; j0 < SP; j0 += 8) {
        const int j = j0 + (lane >> 3), d0 = (lane & 7) * 8;
        half8_t v8 = {0, 0, 0, 0, 0, 0, 0, 0};
        if (j < S) v8 = *reinterpret_cast<const half8_t*>(vb + (long long)j * ld + d0);
#pragma unroll
        for (int e = 0; e < 8; ++e) vt[(d0 + e) * VP + j] = v8[e];
    }
    // K fragments of every key tile (A operand of the score product): rows j = 32 jt + r, d = 16 ks + 8 hh .. +7
    half8_t kf[NT][4];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        const int j = 32 * jt + r;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            kf[jt][ks] = j < S ? *reinterpret_cast<const half8_t*>(kb + (long long)j * ld + 16 * ks + 8 * hh) : z;
        }
    }
    __builtin_amdgcn_s_waitcnt(0);                              // this wave's LDS writes of V^T are done (in-order per wave)
#pragma unroll 1
    for (int it = 0; it < NT; ++it) {
        const int i = 32 * it + r;                             // this lane's query
        if (32 * it >= S) break;
        half8_t qf[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
            qf[ks] = i < S ? *reinterpret_cast<const half8_t*>(qb + (long long)i * ld + 16 * ks + 8 * hh) : z;
        }
        float16_t sc[NT];
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
#pragma unroll
            for (int e = 0; e < 16; ++e) sc[jt][e] = 0.f;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) sc[jt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[jt][ks], qf[ks], sc[jt], 0, 0, 0);
        }
        // scale, mask, softmax over the keys of query i (registers of this lane and of lane ^ 32)
        float m = -INFINITY;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int j = 32 * jt + 8 * (e >> 2) + 4 * hh + (e & 3);
                const bool ok = j < S && !(causal && j > i);
                const float v = ok ? sc[jt][e] * 0.125f : -INFINITY;
                sc[jt][e] = v;
                m = fmaxf(m, v);
            }
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        float sum = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float pe = (sc[jt][e] == -INFINITY) ? 0.f : __expf(sc[jt][e] - m);
                sc[jt][e] = pe;
                sum += pe;
            }
        sum += __shfl_xor(sum, 32, 64);
        const float inv = sum > 0.f ? 1.0f / sum : 0.f;        // (padding queries i >= S: all masked)
        // P^T as B operand: K-slice (jt, q) = registers 8 q .. 8 q + 7 of tile jt  (j = 32 jt + 16 q + {4 hh + k', 8 + 4 hh + k'})
        float16_t oc[2];
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int e = 0; e < 16; ++e) oc[dt][e] = 0.f;
#pragma unroll
        for (int jt = 0; jt < NT; ++jt)
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                half8_t pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (_Float16)(sc[jt][8 * q + e] * inv);
#pragma unroll
                for (int dt = 0; dt < 2; ++dt) {
                    // V^T rows d = 32 dt + r, the same 8 keys: two runs of four consecutive j
                    const _Float16* vr = vt + (32 * dt + r) * VP + 32 * jt + 16 * q + 4 * hh;
                    const half4_t lo4 = *reinterpret_cast<const half4_t*>(vr);
                    const half4_t hi4 = *reinterpret_cast<const half4_t*>(vr + 8);
                    half8_t vf;
#pragma unroll
                    for (int e = 0; e < 4; ++e) { vf[e] = lo4[e]; vf[4 + e] = hi4[e]; }
                    oc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, oc[dt], 0, 0, 0);
                }
            }
        // O^T[d][i]: this lane's query i, d = 32 dt + 8 g + 4 hh + k'
        if (i < S) {
            _Float16* orow = out + ((long long)b * S + i) * E + h * 64;
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    half4_t o4;
#pragma unroll
                    for (int k2 = 0; k2 < 4; ++k2) o4[k2] = (_Float16)oc[dt][4 * g + k2];
                    *reinterpret_cast<half4_t*>(orow + 32 * dt + 8 * g + 4 * hh) = o4;
                }
        }
    }
}

// image (B, 3, R, R) fp16 NCHW -> patch matrix [B * (R/P)^2][3 * P * P], column = c * P*P + ky * P + kx (= the
// flattening of conv1.weight (width, 3, P, P)), so that the patch embedding is one GEMM
__global__ __launch_bounds__(256) void patchify_f16_kernel(const _Float16* __restrict__ img, int B, int R, int P,
                                                           _Float16* __restrict__ out) {
    const int G = R / P, K = 3 * P * P;
    const long long total = (long long)B * G * G * K / 8;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const long long el = e * 8;
        const int col = (int)(el % K);
        const long long row = el / K;
        const int c = col / (P * P), ky = (col / P) % P, kx = col % P;          // kx multiple of 8
        const int b = (int)(row / (G * G)), gy = (int)(row / G % G), gx = (int)(row % G);
        const _Float16* src = img + (((long long)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx;
        *reinterpret_cast<half8_t*>(out + el) = *reinterpret_cast<const half8_t*>(src);
    }
}

// x[b][0] = class_emb + pos[0];  x[b][1 + p] = patch_out[b * NP + p] + pos[1 + p]          (width W)
__global__ __launch_bounds__(256) void vit_tokens_f16_kernel(const _Float16* __restrict__ patch_out, const _Float16* __restrict__ cls,
                                                             const _Float16* __restrict__ pos, int B, int NP, int W,
                                                             _Float16* __restrict__ x) {
    const long long total = (long long)B * (NP + 1) * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % W);
        const long long t = e / W;
        const int s = (int)(t % (NP + 1));
        const long long b = t / (NP + 1);
        const float base = s == 0 ? (float)cls[c] : (float)patch_out[(b * NP + s - 1) * W + c];
        x[e] = (_Float16)(base + (float)pos[(long long)s * W + c]);
    }
}

// x[b][s] = table[tokens[b][s]] + pos[s]
__global__ __launch_bounds__(256) void text_tokens_f16_kernel(const int* __restrict__ tokens, const _Float16* __restrict__ table,
                                                              const _Float16* __restrict__ pos, long long rows, int S, int W,
                                                              _Float16* __restrict__ x) {
    const long long total = rows * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % W);
        const long long t = e / W;
        const int s = (int)(t % S);
        x[e] = (_Float16)((float)table[(long long)tokens[t] * W + c] + (float)pos[(long long)s * W + c]);
    }
}

// out[i] = x[index[i]]  (rows of width W; the class token of every image / the end-of-text token of every caption)
__global__ __launch_bounds__(256) void gather_rows_f16_kernel(const _Float16* __restrict__ x, const long long* __restrict__ index,
                                                              long long n, int W, _Float16* __restrict__ out) {
    const long long total = n * W;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < total; e += (long long)gridDim.x * 256) {
        const int c = (int)(e % W);
        const long long i = e / W;
        out[e] = x[index[i] * W + c];
    }
}

inline int grid1d(long long total) {
    long long b = (total + 255) / 256;
    return (int)(b > 16384 ? 16384 : (b < 1 ? 1 : b));
}

}  // namespace

extern "C" {

int tise_gemm_f16(const void* a_dev, int64_t lda, const void* w_dev, int64_t ldw, const void* bias_dev, const void* res_dev,
                  int64_t ldr, void* out_dev, int64_t ldo, int m, int n, int k, int act, void* stream) {
    if (!a_dev || !w_dev || !out_dev || m < 0 || n <= 0 || k <= 0 || k % GM_BK != 0 || n % 8 != 0 || lda % 8 != 0 || ldw % 8 != 0 ||
        ldo % 8 != 0 || (res_dev && ldr % 8 != 0) || (act != 0 && act != 1))
        return TISE_ERR_INVALID_ARG;
    if (m == 0) return TISE_OK;
    GemmArgs p;
    p.a = reinterpret_cast<const _Float16*>(a_dev); p.lda = lda;
    p.w = reinterpret_cast<const _Float16*>(w_dev); p.ldw = ldw;
    p.bias = reinterpret_cast<const _Float16*>(bias_dev);
    p.res = reinterpret_cast<const _Float16*>(res_dev); p.ldr = ldr;
    p.out = reinterpret_cast<_Float16*>(out_dev); p.ldo = ldo;
    p.M = m; p.N = n; p.K = k; p.act = act;
    const long long tiles = (long long)((m + 127) / 128) * ((n + 127) / 128);
    if (tiles > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    // TISE_GEMM_BIG: 0 never, 1 (default) by the rule below, 2 always -- the A/B switch of tools/clip_gemm_probe.py
    static const int big_mode = [] { const char* e = getenv("TISE_GEMM_BIG"); return e ? atoi(e) : 1; }();
    const long long tiles_big = (long long)((m + 255) / 256) * ((n + 255) / 256);
    // TISE_GEMM_SHAPE: 16 = the v_mfma_f32_16x16x32_f16 kernels for every launch, 32 = round 2's 32x32x16 kernels for every
    // launch, unset = by measurement (tools/clip_gemm_probe.py, profiles/r04b_clip_gemm_shapes.txt): the 128 x 128 kernel
    // is 12-14 % faster on the K = 32 shape (660 -> 754, 889 -> 1001, 877 -> 984 TFLOP/s), the 256 x 256 kernel 8-14 %
    // SLOWER (its 128 x 64 wave tile needs the slice-1 fragments re-loaded row block by row block to stay inside 256
    // registers, which serialises its issue stream), so it stays on 32x32x16
    static const int shape = [] { const char* e = getenv("TISE_GEMM_SHAPE"); return e ? atoi(e) : 0; }();
    const bool big = big_mode == 2 || (big_mode == 1 && tiles_big >= 768);
    if (shape == 16 || (shape == 0 && !big)) {
        constexpr int lds_big = 2 * (256 * 128 + 256 * 128), lds_small = 2 * (128 * 128 + 128 * 128);
        static std::atomic<unsigned long long> attr16{0};
        if (tise_first_use_on_this_device(attr16)) {
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm16_f16_kernel<true>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm16_f16_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_small));
        }
        if (big) hipLaunchKernelGGL(gemm16_f16_kernel<true>, dim3((unsigned)tiles_big), dim3(512), lds_big, (hipStream_t)stream, p);
        else hipLaunchKernelGGL(gemm16_f16_kernel<false>, dim3((unsigned)tiles), dim3(256), lds_small, (hipStream_t)stream, p);
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    if (big) {
        constexpr int lds_big = 2 * (256 * 128 + 256 * 128);
        static std::atomic<unsigned long long> attr_set{0};
        if (tise_first_use_on_this_device(attr_set))
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_big_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_big));
        hipLaunchKernelGGL(gemm_f16_big_kernel, dim3((unsigned)tiles_big), dim3(512), lds_big, (hipStream_t)stream, p);
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    hipLaunchKernelGGL(gemm_f16_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_layernorm_f16(const void* x_dev, int64_t ldx, const void* gamma_dev, const void* beta_dev, void* out_dev, int64_t ldo,
                       int64_t rows, int C, float eps, void* stream) {
    if (!x_dev || !gamma_dev || !beta_dev || !out_dev || rows < 0 || C <= 0 || C > 1024 || C % 8 != 0 || ldx % 8 != 0 || ldo % 8 != 0 ||
        ((reinterpret_cast<uintptr_t>(x_dev) | reinterpret_cast<uintptr_t>(out_dev) | reinterpret_cast<uintptr_t>(gamma_dev) |
          reinterpret_cast<uintptr_t>(beta_dev)) & 15) != 0)
        return TISE_ERR_INVALID_ARG;
    if (rows == 0) return TISE_OK;
    if ((rows + 3) / 4 > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(layernorm_f16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x_dev), ldx, reinterpret_cast<const _Float16*>(gamma_dev),
                       reinterpret_cast<const _Float16*>(beta_dev), reinterpret_cast<_Float16*>(out_dev), ldo, (int)rows, C, eps);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_attention_f16(const void* qkv_dev, int batch, int seq, int heads, int head_dim, int causal, void* out_dev, void* stream) {
    if (!qkv_dev || !out_dev || batch < 0 || seq <= 0 || seq > 96 || heads <= 0 || head_dim != 64) return TISE_ERR_INVALID_ARG;
    if (batch == 0) return TISE_OK;
    const long long bh = (long long)batch * heads;
    if ((bh + 3) / 4 > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)((bh + 3) / 4)), block(256);
    const _Float16* q = reinterpret_cast<const _Float16*>(qkv_dev);
    _Float16* o = reinterpret_cast<_Float16*>(out_dev);
    if (seq <= 32) hipLaunchKernelGGL(attention_f16_kernel<1>, grid, block, 0, (hipStream_t)stream, q, seq, heads, (int)bh, causal, o);
    else if (seq <= 64) hipLaunchKernelGGL(attention_f16_kernel<2>, grid, block, 0, (hipStream_t)stream, q, seq, heads, (int)bh, causal, o);
    else hipLaunchKernelGGL(attention_f16_kernel<3>, grid, block, 0, (hipStream_t)stream, q, seq, heads, (int)bh, causal, o);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_patchify_f16(const void* img_dev, int batch, int res, int patch, void* out_dev, void* stream) {
    if (!img_dev || !out_dev || batch < 0 || res <= 0 || patch <= 0 || res % patch != 0 || patch % 8 != 0) return TISE_ERR_INVALID_ARG;
    if (batch == 0) return TISE_OK;
    const long long total = (long long)batch * (res / patch) * (res / patch) * 3 * patch * patch / 8;
    hipLaunchKernelGGL(patchify_f16_kernel, dim3(grid1d(total)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(img_dev), batch, res, patch, reinterpret_cast<_Float16*>(out_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_vit_tokens_f16(const void* patch_out_dev, const void* class_emb_dev, const void* pos_emb_dev, int batch, int n_patches,
                        int width, void* x_dev, void* stream) {
    if (!patch_out_dev || !class_emb_dev || !pos_emb_dev || !x_dev || batch < 0 || n_patches <= 0 || width <= 0) return TISE_ERR_INVALID_ARG;
    if (batch == 0) return TISE_OK;
    hipLaunchKernelGGL(vit_tokens_f16_kernel, dim3(grid1d((long long)batch * (n_patches + 1) * width)), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const _Float16*>(patch_out_dev), reinterpret_cast<const _Float16*>(class_emb_dev),
                       reinterpret_cast<const _Float16*>(pos_emb_dev), batch, n_patches, width, reinterpret_cast<_Float16*>(x_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_text_tokens_f16(const int32_t* tokens_dev, const void* table_dev, const void* pos_emb_dev, int64_t rows, int seq, int width,
                         void* x_dev, void* stream) {
    if (!tokens_dev || !table_dev || !pos_emb_dev || !x_dev || rows < 0 || seq <= 0 || width <= 0) return TISE_ERR_INVALID_ARG;
    if (rows == 0) return TISE_OK;
    hipLaunchKernelGGL(text_tokens_f16_kernel, dim3(grid1d(rows * width)), dim3(256), 0, (hipStream_t)stream, tokens_dev,
                       reinterpret_cast<const _Float16*>(table_dev), reinterpret_cast<const _Float16*>(pos_emb_dev), (long long)rows, seq,
                       width, reinterpret_cast<_Float16*>(x_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_gather_rows_f16(const void* x_dev, const int64_t* index_dev, int64_t n, int width, void* out_dev, void* stream) {
    if (!x_dev || !index_dev || !out_dev || n < 0 || width <= 0) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    hipLaunchKernelGGL(gather_rows_f16_kernel, dim3(grid1d(n * width)), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x_dev), reinterpret_cast<const long long*>(index_dev), (long long)n, width,
                       reinterpret_cast<_Float16*>(out_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // extern "C"
