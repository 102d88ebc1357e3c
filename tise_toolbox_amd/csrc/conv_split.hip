// Implicit-GEMM convolution on fp16 MFMA with 3-term split-precision operands (fp32-class accuracy).
//
// Why: the InceptionV3 convolutions are 86 % of a step.  MIOpen's fp32 kernels already run at 60-85 % of
// the fp32 MFMA peak (155 TFLOP/s; profiles/r01*_conv_probe); gfx950 has no TF32/xf32, so the only faster
// matrix path is the 16x faster fp16/bf16 MFMA.  Plain fp16 is far outside the |dFID| <= 1e-3 budget, so
// every fp32 value v is carried as TWO fp16 numbers
//        v  ~=  hi + lo * 2^-11,      hi = fp16(v),   lo = fp16((v - hi) * 2^11)           (22 mantissa bits)
// and a product is evaluated with three MFMAs, accumulated in fp32:
//        a*b ~= a_hi*b_hi + 2^-11 (a_hi*b_lo + a_lo*b_hi)          (the dropped lo*lo term is 2^-22 relative)
// i.e. 3/16 of the fp32-MFMA cost per product at ~2^-22 relative accuracy per term.
//
// Activations live in HBM in that split form (two fp16 planes, NHWC) -- the same 4 bytes per element as
// fp32 -- so the hot loop loads fp16 fragments directly and never converts.  The epilogue applies the
// folded-BatchNorm scale/bias and ReLU, re-splits the fp32 accumulators and writes straight into a
// channel slice of the consumer's tensor (block concat buffer / next conv input), or writes raw fp32
// for the pool branch (whose 3x3 average runs after the 1x1 conv, trunk_ops.hip).
//
// GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin (cin fastest; weights pre-packed
// [plane][Cout_pad][K_pad] with K contiguous, so A rows and B rows are both K-contiguous 16-byte fragments).
// Workgroup = 256 threads = 4 waves stacked along M; tile = 128 pixels x (32*TN) couts x 32 k per step;
// each wave owns 32 x 32*TN: TN accumulator pairs (main, corr) of v_mfma_f32_32x32x16_f16.
// Operands are staged global -> registers -> LDS (rows padded to 80 B: conflict-free ds_read_b128, see
// MI355X_MICROARCH LDS table) with the next K-slab's loads in flight under the current slab's MFMAs;
// out-of-image taps and M/N/K tails are zero-filled in registers (that is why the A operand is not
// fetched with global_load_lds).
#include <hip/hip_fp16.h>
#include "common.h"
#include "conv_epilogue.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // native vector: stays in registers (SROA)

#define CS_BM 128
#define CS_BK 32
#define CS_PITCH 80                          // bytes per LDS row (64 B of data + 16 B pad)
#define CS_A_PLANE (CS_BM * CS_PITCH)        // 10240 B

typedef tise_conv_seg ConvSeg;
typedef tise_conv_args ConvArgs;

// Tile row -> output pixel index (or -1).  GRID = false: tile rows ARE output pixels.  GRID = true (window
// kernel): tile rows are pixels of the INPUT grid (n, y, x); the output pixel is (n, y, x) when it exists.
template <bool GRID>
__device__ __forceinline__ long long conv_out_pixel(const ConvArgs& p, long long g) {
    if (!GRID) return g < p.M ? g : -1;
    const long long hw = (long long)p.H * p.W;
    if (g >= (long long)p.N * hw) return -1;
    const long long n = g / hw;
    const int rem = (int)(g - n * hw);
    const int y = rem / p.W, x = rem - y * p.W;
    if (y >= p.OH || x >= p.OW) return -1;
    return (n * p.OH + y) * p.OW + x;
}

// Epilogue shared by all kernels (see the comment inside).  LDS_BYTES is the size of the caller's LDS array.
template <int TN, int LDS_BYTES, int BM = CS_BM, bool GRID = false>
__device__ __forceinline__ void conv_split_epilogue(const ConvArgs& p, float16_t (&acc_main)[TN], float16_t (&acc_corr)[TN],
                                                    unsigned char* lds, long long m0, int n0) {
    constexpr int BN = 32 * TN;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- epilogue ---------------------------------------------------------------------------------
    // D[row = pixel][col = cout]; a lane holds col = lane & 31 and rows (j&3) + 8*(j>>2) + 4*(lane>>5).
    // Split outputs go through LDS (the operand buffers are free now) so that every pixel row of the tile
    // leaves as 16-byte stores covering whole 64..320-byte channel runs; one pass per plane.  Raw fp32
    // segments (pool branch) are stored directly from the accumulators.
    constexpr int T_PITCH = BN * 2 + 16;                    // bytes per staged pixel row
    static_assert(BM * T_PITCH <= LDS_BYTES, "staging tile must fit the operand LDS");
    constexpr int NCH = BN / 8;                             // 16-byte chunks per row
    const int nseg = p.nseg & 0xff;
    _Float16 lo_keep[TN][16];
    float vmax = 0.f;                                        // range guard of the split format (common.h)
#pragma unroll
    for (int plane = 0; plane < 2; ++plane) {
#pragma unroll
        for (int t = 0; t < TN; ++t) {
            const int col = n0 + t * 32 + (lane & 31);
            const bool col_ok = col < p.Cout;
            // static-index segment look-up (dynamic indexing of the by-value argument struct would spill it)
            void* s_dst = p.seg[0].dst;
            long long s_ld = p.seg[0].ld;
            int s_off = p.seg[0].off, s_mode = p.seg[0].mode, s_c0 = p.seg[0].c0;
#pragma unroll
            for (int s = 1; s < 4; ++s)
                if (s < nseg && col >= p.seg[s].c0) {
                    s_dst = p.seg[s].dst; s_ld = p.seg[s].ld; s_off = p.seg[s].off; s_mode = p.seg[s].mode; s_c0 = p.seg[s].c0;
                }
            const float sc = col_ok ? p.scale[col] : 0.f;
            const float bs = col_ok ? p.bias[col] : 0.f;
            unsigned char* trow = lds + (size_t)(wave * 32 + 4 * (lane >> 5)) * T_PITCH + (t * 32 + (lane & 31)) * 2;
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int r = (j & 3) + 8 * (j >> 2);
                if (plane == 0) {
                    float v = (acc_main[t][j] + acc_corr[t][j] * (1.0f / 2048.0f)) * sc;
                    if (s_mode == 0) {
                        v = fmaxf(v + bs, 0.f);
                        vmax = fmaxf(vmax, v);
                        const _Float16 hi = (_Float16)v;
                        lo_keep[t][j] = (_Float16)((v - (float)hi) * 2048.0f);
                        *reinterpret_cast<_Float16*>(trow + r * T_PITCH) = hi;
                    } else {
                        lo_keep[t][j] = (_Float16)0.f;
                        const long long pp = conv_out_pixel<GRID>(p, m0 + wave * 32 + r + 4 * (lane >> 5));
                        if (col_ok && pp >= 0) reinterpret_cast<float*>(s_dst)[pp * s_ld + s_off + (col - s_c0)] = v;
                    }
                } else {
                    *reinterpret_cast<_Float16*>(trow + r * T_PITCH) = lo_keep[t][j];
                }
            }
        }
        __syncthreads();
        for (int idx = tid; idx < BM * NCH; idx += BM * 2) {
            const int r = idx / NCH, c = idx - r * NCH;
            const int col = n0 + c * 8;
            if (col >= p.Cout) continue;
            const long long pp = conv_out_pixel<GRID>(p, m0 + r);
            if (pp < 0) continue;
            void* s_dst = p.seg[0].dst;
            long long s_ld = p.seg[0].ld, s_plane = p.seg[0].plane;
            int s_off = p.seg[0].off, s_mode = p.seg[0].mode, s_c0 = p.seg[0].c0;
#pragma unroll
            for (int s = 1; s < 4; ++s)
                if (s < nseg && col >= p.seg[s].c0) {
                    s_dst = p.seg[s].dst; s_ld = p.seg[s].ld; s_plane = p.seg[s].plane;
                    s_off = p.seg[s].off; s_mode = p.seg[s].mode; s_c0 = p.seg[s].c0;
                }
            if (s_mode != 0) continue;
            const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lds + (size_t)r * T_PITCH + c * 16);
            _Float16* d = reinterpret_cast<_Float16*>(s_dst) + (plane ? s_plane : 0) + pp * s_ld + s_off + (col - s_c0);
            *reinterpret_cast<u32x4_t*>(d) = v;
        }
        if (plane == 0) __syncthreads();
    }
    tise_flag_split_overflow(vmax);
}

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_split_kernel(const ConvArgs p) {
    constexpr int BN = 32 * TN;
    constexpr int B_PLANE = BN * CS_PITCH;
    constexpr int B_ITERS = (BN * 2 + 255) / 256;
    __shared__ __attribute__((aligned(16))) unsigned char lds[2 * CS_A_PLANE + 2 * B_PLANE];
    unsigned char* As = lds;
    unsigned char* Bs = lds + 2 * CS_A_PLANE;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware remap: consecutive logical tiles (same pixel rows, neighbouring cout tiles) share an L2
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const long long nwg = (long long)gridDim.x;
    long long bid = blockIdx.x;
    {
        const long long q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long tile_m = bid / tiles_n;
    const int tile_n = (int)(bid - tile_m * tiles_n);
    const long long m0 = tile_m * CS_BM;
    const int n0 = tile_n * BN;

    // ---- A loader role: one (row, 16-channel unit) per thread ---------------------------------------
    const int arow = tid >> 1, aunit = tid & 1;
    const long long pix = m0 + arow;
    const bool row_ok = pix < p.M;
    int ih0, iw0;
    long long img_base;
    {
        const long long pp = row_ok ? pix : 0;
        const int ohw = p.OH * p.OW;
        const int n = (int)(pp / ohw);
        const int rem = (int)(pp - (long long)n * ohw);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        ih0 = oh * p.SH - p.PH;
        iw0 = ow * p.SW - p.PW;
        img_base = (long long)n * p.H * p.W * p.Cin;
    }
    int a_c = aunit * 16, a_kh = 0, a_kw = 0;           // running (kh, kw, c) of this thread's unit
    while (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }
    int a_k = aunit * 16;                               // running k index (for the K tail)

    u32x4_t ra[4];                                        // hi: 2 x 16 B, lo: 2 x 16 B
    u32x4_t rb[B_ITERS][4];

#define CS_FETCH(STEP)                                                                                  \
    {                                                                                                   \
        const int ih = ih0 + a_kh, iw = iw0 + a_kw;                                                      \
        const bool ok = row_ok && a_k < p.K && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;               \
        const long long off = ok ? img_base + ((long long)ih * p.W + iw) * p.Cin + a_c : 0;              \
        const u32x4_t* gh = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const _Float16*>(p.x) + off);                                     \
        const u32x4_t* gl = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const _Float16*>(p.x) + p.x_plane + off);                         \
        const unsigned mk = ok ? 0xffffffffu : 0u;     /* mask by value: a select of loads goes to scratch */ \
        ra[0] = gh[0]; ra[1] = gh[1]; ra[2] = gl[0]; ra[3] = gl[1];                                      \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) ra[q] &= mk;                                        \
        _Pragma("unroll") for (int it = 0; it < B_ITERS; ++it) {                                         \
            int idx = tid + 256 * it;                  /* unconditional loads (clamped), conditional LDS stores */ \
            idx = idx < BN * 2 ? idx : BN * 2 - 1;                                                       \
            const long long wo = (long long)(n0 + (idx >> 1)) * p.Kpad + (STEP) * CS_BK + (idx & 1) * 16; \
            const u32x4_t* wh = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const _Float16*>(p.w) + wo);                                  \
            const u32x4_t* wl = reinterpret_cast<const u32x4_t*>(reinterpret_cast<const _Float16*>(p.w) + p.w_plane + wo);                      \
            rb[it][0] = wh[0]; rb[it][1] = wh[1]; rb[it][2] = wl[0]; rb[it][3] = wl[1];                  \
        }                                                                                                \
        a_k += CS_BK; a_c += CS_BK;                                                                      \
        if (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }                    \
        if (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }                    \
    }

    float16_t acc_main[TN], acc_corr[TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc_main[t][j] = 0.f; acc_corr[t][j] = 0.f; }

    const int nsteps = p.Kpad / CS_BK;
    CS_FETCH(0)
    const int frag_off = (lane & 31) * CS_PITCH + (lane >> 5) * 16;
    for (int step = 0; step < nsteps; ++step) {
        // registers -> LDS
        {
            unsigned char* d = As + arow * CS_PITCH + aunit * 32;
            *reinterpret_cast<u32x4_t*>(d) = ra[0];
            *reinterpret_cast<u32x4_t*>(d + 16) = ra[1];
            *reinterpret_cast<u32x4_t*>(d + CS_A_PLANE) = ra[2];
            *reinterpret_cast<u32x4_t*>(d + CS_A_PLANE + 16) = ra[3];
#pragma unroll
            for (int it = 0; it < B_ITERS; ++it) {
                const int idx = tid + 256 * it;
                if (idx < BN * 2) {
                    unsigned char* e = Bs + (idx >> 1) * CS_PITCH + (idx & 1) * 32;
                    *reinterpret_cast<u32x4_t*>(e) = rb[it][0];
                    *reinterpret_cast<u32x4_t*>(e + 16) = rb[it][1];
                    *reinterpret_cast<u32x4_t*>(e + B_PLANE) = rb[it][2];
                    *reinterpret_cast<u32x4_t*>(e + B_PLANE + 16) = rb[it][3];
                }
            }
        }
        __syncthreads();
        if (step + 1 < nsteps && !(p.nseg & 0x100)) CS_FETCH(step + 1)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned char* ap = As + wave * 32 * CS_PITCH + frag_off + s * 32;
            const half8_t a_hi = *reinterpret_cast<const half8_t*>(ap);
            const half8_t a_lo = *reinterpret_cast<const half8_t*>(ap + CS_A_PLANE);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const unsigned char* bp = Bs + t * 32 * CS_PITCH + frag_off + s * 32;
                const half8_t b_hi = *reinterpret_cast<const half8_t*>(bp);
                const half8_t b_lo = *reinterpret_cast<const half8_t*>(bp + B_PLANE);
                acc_main[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_hi, acc_main[t], 0, 0, 0);
                acc_corr[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_hi, b_lo, acc_corr[t], 0, 0, 0);
                acc_corr[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_lo, b_hi, acc_corr[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    conv_split_epilogue<TN, 2 * CS_A_PLANE + 2 * B_PLANE>(p, acc_main, acc_corr, lds, m0, n0);
}

// ------------------------------------------------------------------------------------------------
// Variant 2: operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers), two LDS
// stages, one barrier per K-step: the loads of step s+1 are issued right after the barrier of step s and
// stay in flight under that step's MFMAs.  LDS rows are unpadded 64-byte runs (a DMA wave-instruction
// writes 1 KiB = 16 rows linearly); bank conflicts of the ds_read_b128 fragment reads are removed by an
// XOR swizzle of the 16-byte chunk index with (row >> 2) & 3, applied on the SOURCE address of the DMA and
// on the read address (cdna guide rule 21).  Out-of-image taps and tails read a zero page instead.
__device__ u32x4_t g_conv_zero_page[4];

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int TN>
__global__ __launch_bounds__(256, 2) void conv_split_glds_kernel(const ConvArgs p) {
    constexpr int BN = 32 * TN;
    constexpr int A_PLANE = CS_BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int T_BYTES = CS_BM * (BN * 2 + 16);
    constexpr int LDS_BYTES = (2 * STAGE > T_BYTES) ? 2 * STAGE : T_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.Cout + BN - 1) / BN;
    const long long nwg = (long long)gridDim.x;
    long long bid = blockIdx.x;
    {
        const long long q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const long long tile_m = bid / tiles_n;
    const int tile_n = (int)(bid - tile_m * tiles_n);
    const long long m0 = tile_m * CS_BM;
    const int n0 = tile_n * BN;

    // ---- loader role -------------------------------------------------------------------------------
    // lane i of a DMA instruction fills row (i >> 2), physical chunk (i & 3) of a 16-row block; the logical
    // chunk it must fetch is (i & 3) ^ ((row >> 2) & 3) = (i & 3) ^ ((i >> 4) & 3)
    const int cl = (lane & 3) ^ ((lane >> 4) & 3);
    const int unit = cl >> 1, sub8 = (cl & 1) * 8;
    const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
    const _Float16* wg = reinterpret_cast<const _Float16*>(p.w);
    const _Float16* zp = reinterpret_cast<const _Float16*>(g_conv_zero_page);
    int ih0[2], iw0[2];
    long long ibase[2];
    bool rok[2];
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const long long pix = m0 + (2 * wave + jj) * 16 + (lane >> 2);
        rok[jj] = pix < p.M;
        const long long pp = rok[jj] ? pix : 0;
        const int ohw = p.OH * p.OW;
        const int n = (int)(pp / ohw);
        const int rem = (int)(pp - (long long)n * ohw);
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        ih0[jj] = oh * p.SH - p.PH;
        iw0[jj] = ow * p.SW - p.PW;
        ibase[jj] = (long long)n * p.H * p.W * p.Cin;
    }
    int a_c = unit * 16, a_kh = 0, a_kw = 0, a_k = unit * 16;
    while (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }

#define CG_ISSUE(STEP, STAGEBASE)                                                                         \
    {                                                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                 \
            const int ih = ih0[jj] + a_kh, iw = iw0[jj] + a_kw;                                            \
            const bool ok = rok[jj] && a_k < p.K && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;            \
            const _Float16* sh = xg + ibase[jj] + ((long long)ih * p.W + iw) * p.Cin + a_c + sub8;         \
            const _Float16* sl = sh + p.x_plane;                                                           \
            sh = ok ? sh : zp; sl = ok ? sl : zp;                                                          \
            unsigned char* d = (STAGEBASE) + (2 * wave + jj) * 1024;                                       \
            __builtin_amdgcn_global_load_lds(sh, (lds_ptr_t)d, 16, 0, 0);                                  \
            __builtin_amdgcn_global_load_lds(sl, (lds_ptr_t)(d + A_PLANE), 16, 0, 0);                      \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                   \
            const int q = wave * TN + i;                      /* 4*TN instructions: 2 planes x 2*TN blocks */ \
            const int plane = q >= 2 * TN ? 1 : 0;                                                         \
            const int rb = q - plane * 2 * TN;                                                             \
            const _Float16* sw = wg + (plane ? p.w_plane : 0) + (long long)(n0 + rb * 16 + (lane >> 2)) * p.Kpad + \
                                 (STEP) * CS_BK + cl * 8;                                                  \
            unsigned char* d = (STAGEBASE) + 2 * A_PLANE + plane * B_PLANE + rb * 1024;                    \
            __builtin_amdgcn_global_load_lds(sw, (lds_ptr_t)d, 16, 0, 0);                                  \
        }                                                                                                  \
        a_k += CS_BK; a_c += CS_BK;                                                                        \
        if (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }                      \
        if (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }                      \
    }

    float16_t acc_main[1][TN], acc_corr[1][TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc_main[0][t][j] = 0.f; acc_corr[0][t][j] = 0.f; }
    // scale / bias of this tile's couts, 4 per thread, fetched now so that the epilogue never waits on global memory
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }

    const int nsteps = p.Kpad / CS_BK;
    // fragment read: row (lane & 31) of a 32-row block, logical chunk 2*s + (lane >> 5), swizzled
    const int frow = (lane & 31) * 64;
    const int fswz = ((lane & 31) >> 2) & 3;
    CG_ISSUE(0, lds)
    for (int step = 0; step < nsteps; ++step) {
        unsigned char* cur = lds + (step & 1) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my DMA pieces of this stage have landed
        __syncthreads();                                       // everyone's have; everyone left the other stage
        if (step + 1 < nsteps) CG_ISSUE(step + 1, lds + ((step + 1) & 1) * STAGE)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int choff = ((2 * s + (lane >> 5)) ^ fswz) * 16;
            const unsigned char* ap = cur + wave * 32 * 64 + frow + choff;
            const half8_t a_hi = *reinterpret_cast<const half8_t*>(ap);
            const half8_t a_lo = *reinterpret_cast<const half8_t*>(ap + A_PLANE);
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const unsigned char* bp = cur + 2 * A_PLANE + t * 32 * 64 + frow + choff;
                const half8_t b_hi = *reinterpret_cast<const half8_t*>(bp);
                const half8_t b_lo = *reinterpret_cast<const half8_t*>(bp + B_PLANE);
                // weight fragment first: the accumulator is D[cout][pixel] (conv_epilogue.h)
                acc_main[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_hi, acc_main[0][t], 0, 0, 0);
                acc_corr[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_lo, a_hi, acc_corr[0][t], 0, 0, 0);
                acc_corr[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b_hi, a_lo, acc_corr[0][t], 0, 0, 0);
            }
        }
    }
    __syncthreads();                                           // all fragment reads done before LDS is reused
    constexpr int ETW = TN > 1 ? 2 : 1;                       // accumulator tiles staged together: 128-byte runs
    constexpr int EPI0 = 4 * conv_epi::Staging<ETW>::BYTES;   // scale / bias / chunk descriptors behind the staging tiles
    static_assert(EPI0 + conv_epi::EpiArea<BN>::BYTES <= LDS_BYTES, "epilogue staging must fit the operand LDS");
    conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre);
    __syncthreads();
    conv_epi::store_tiles_desc<TN, ETW>(p, acc_main, acc_corr, lds + wave * conv_epi::Staging<ETW>::BYTES, lds + EPI0, m0 + wave * 32);
}

// ------------------------------------------------------------------------------------------------
// Variant 5 ("fast", Cin % 32 == 0): variant 2 with the address arithmetic taken out of the K loop.
// rocprofv3 counters on variant 2 (SQ_INSTS_VALU / SQ_VALU_MFMA_BUSY_CYCLES) showed ~130 vector and ~55
// scalar instructions per wave per K-step next to 24 MFMAs: the per-step recomputation of every DMA source
// address (64-bit multiplies, tap decode, bounds checks) cost about as many issue cycles as the MFMAs.
// With Cin a multiple of 32 a K-step never straddles a filter tap, so the K loop is (tap, channel block):
// source pointers are rebuilt only when the tap changes (wave-uniform branch) and otherwise advance by
// 64 bytes; weight pointers always advance by 64 bytes; the two LDS stages are addressed with compile-time
// offsets (loop unrolled by two).
template <int TN, bool DBG = false>
__global__ __launch_bounds__(256, 2) void conv_split_fast_kernel(const ConvArgs p) {
    constexpr int BN = 32 * TN;
    constexpr int A_PLANE = CS_BM * 64, B_PLANE = BN * 64;
    constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;
    constexpr int T_BYTES = CS_BM * (BN * 2 + 16);
    constexpr int LDS_BYTES = (2 * STAGE > T_BYTES) ? 2 * STAGE : T_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS / DMA addresses stay on the SALU
    // 32-bit index arithmetic throughout the prologue (the launcher sends M >= 2^31 to the glds kernel): the 64-bit
    // divisions of the generic kernels cost several hundred instructions per workgroup, ~15 % of a 27-step tile
    const unsigned tiles_n = (unsigned)(p.Cout + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned tile_m = bid / tiles_n;
    const int tile_n = (int)(bid - tile_m * tiles_n);
    const long long m0 = (long long)tile_m * CS_BM;
    const int n0 = tile_n * BN;

    const int cl = (lane & 3) ^ ((lane >> 4) & 3);        // logical 16-byte chunk this lane's DMA piece fetches
    const _Float16* xg = reinterpret_cast<const _Float16*>(p.x);
    const _Float16* zp = reinterpret_cast<const _Float16*>(g_conv_zero_page);
    int ih0[2], iw0[2];
    const _Float16* img[2];
    bool rok[2];
    const unsigned ohw = (unsigned)(p.OH * p.OW), M32 = (unsigned)p.M;
#pragma unroll
    for (int jj = 0; jj < 2; ++jj) {
        const unsigned pix = tile_m * CS_BM + (2 * wave + jj) * 16 + (lane >> 2);
        rok[jj] = pix < M32;
        const unsigned pp = rok[jj] ? pix : 0u;
        const unsigned n = pp / ohw;
        const unsigned rem = pp - n * ohw;
        const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
        ih0[jj] = (int)oh * p.SH - p.PH;
        iw0[jj] = (int)ow * p.SW - p.PW;
        img[jj] = xg + (long long)n * p.H * p.W * p.Cin;
    }
    // weight pointers of this wave's TN DMA pieces.  The weights are packed [cout][k / 32][hi 32 | lo 32] for this kernel
    // (conv_split.py): a (cout, K-step) row is ONE 128-byte line, a piece = 8 couts x 128 B -- the LDS-DMA path moves
    // 128-byte rows at 38 B/clk/CU against 30 for the 64-byte rows of the pixel operand (profiles/
    // r01g_glds_rate_microbench.txt), and the weights are half of a 128-cout step's bytes.  LDS rows are 128 B too; the
    // ds_read_b128 fragments stay conflict-free with the chunk index XOR-ed with (row >> 1) & 7 (eight rows of one
    // parity x eight chunk slots, two parities: the 16 lanes of a b128 group hit 16 distinct 16-byte bank groups).
    const _Float16* pb[TN];
    int pb_off[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int q = wave * TN + i;                                   // 8-cout block of the tile
        const int r = q * 8 + (lane >> 3);                             // cout row within the tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);                     // logical chunk: 0-3 hi, 4-7 lo
        pb[i] = reinterpret_cast<const _Float16*>(p.w) + (long long)(n0 + r) * (2 * p.Kpad) + c * 8;
        pb_off[i] = 2 * A_PLANE + q * 1024;
    }
    // pixel-operand pointers for the current tap: ALWAYS loadable (the zero page when the tap falls outside the
    // image or the row outside M) with a per-lane advance of 64 bytes or 0, so that a K-step issues its DMA
    // straight from the registers and spends one 64-bit add per pointer (the null-pointer selects this replaces
    // were 16 of the ~23 vector instructions per step; SQ_INSTS_VALU / SQ_INSTS_MFMA was 3.4-5.7)
    const unsigned char* pa_hi[2];
    const unsigned char* pa_lo[2];
    long long pa_inc[2];                                      // bytes
    // K order.  Phase A: (tap, full 32-channel block), nA = KH*KW*(Cin/32) steps.  Phase B (Cin % 32 == 16 only):
    // the 16-channel tails of the taps, TWO TAPS PER STEP -- chunks 0,1 of a step carry the tail of tap 2j, chunks
    // 2,3 that of tap 2j+1 (lanes pick their tap by their chunk) -- so no MFMA runs on padding except in the last
    // step of an odd tap count.  The weights are packed in the same order (conv_split.py).
    int kh = 0, kw = 0, cblk = 0, istep = 0;
    const int ncblk = p.Cin / CS_BK, ntaps = p.KH * p.KW;
    const int nA = ntaps * ncblk;
    const int nB = (p.Cin & 16) ? (ntaps + 1) / 2 : 0;
#define CF_TAP_AT(KH_, KW_, CH_, VALID_)                                                                  \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                     \
        const int ih = ih0[jj] + (KH_), iw = iw0[jj] + (KW_);                                              \
        const bool ok = rok[jj] && (VALID_) && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;                 \
        const _Float16* src = img[jj] + ((long long)ih * p.W + iw) * p.Cin + (CH_);                        \
        pa_hi[jj] = reinterpret_cast<const unsigned char*>(ok ? src : zp);                                 \
        pa_lo[jj] = reinterpret_cast<const unsigned char*>(ok ? src + p.x_plane : zp);                     \
        pa_inc[jj] = ok ? CS_BK * 2 : 0;                                                                   \
    }
#define CF_TAP() CF_TAP_AT(kh, kw, cl * 8, kh < p.KH)
// phase B step j: this lane's tap is 2j + (cl >> 1), its 8 channels start at 32*ncblk + 8*(cl & 1)
#define CF_TAP_B(J)                                                                                       \
    {                                                                                                     \
        const int tl = 2 * (J) + (cl >> 1);                                                                \
        const int lkh = tl / p.KW, lkw = tl - lkh * p.KW;                                                  \
        CF_TAP_AT(lkh, lkw, ncblk * CS_BK + (cl & 1) * 8, tl < ntaps)                                      \
    }
#define CF_ISSUE(STAGEOFF)                                                                                \
    {                                                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                 \
            /* locals on purpose: with array elements as direct builtin arguments hipcc (ROCm 7.2) silently  \
               drops the host-side launch stub of this template */                                         \
            const unsigned char* sh = pa_hi[jj];                                                           \
            const unsigned char* sl = pa_lo[jj];                                                           \
            __builtin_amdgcn_global_load_lds(sh, (lds_ptr_t)(lds + (STAGEOFF) + (2 * wave + jj) * 1024), 16, 0, 0);            \
            __builtin_amdgcn_global_load_lds(sl, (lds_ptr_t)(lds + (STAGEOFF) + A_PLANE + (2 * wave + jj) * 1024), 16, 0, 0);  \
            pa_hi[jj] = sh + pa_inc[jj];                                                                   \
            pa_lo[jj] = sl + pa_inc[jj];                                                                   \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                   \
            const _Float16* sw_ = pb[i];                                                                   \
            unsigned char* dw_ = lds + (STAGEOFF) + pb_off[i];                                             \
            __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)dw_, 16, 0, 0);                               \
            pb[i] += 2 * CS_BK;                              /* next K-step: 128 bytes further */          \
        }                                                                                                  \
        ++istep;                                             /* the step whose pointers are prepared now */ \
        if (istep >= nA) {                                   /* wave-uniform */                            \
            if (istep < nA + nB) CF_TAP_B(istep - nA)                                                      \
        } else if (++cblk == ncblk) {                        /* next filter tap */                         \
            cblk = 0;                                                                                      \
            if (++kw == p.KW) { kw = 0; ++kh; }                                                            \
            CF_TAP()                                                                                       \
        }                                                                                                  \
    }

    float16_t acc_main[1][TN], acc_corr[1][TN];
#pragma unroll
    for (int t = 0; t < TN; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) { acc_main[0][t][j] = 0.f; acc_corr[0][t][j] = 0.f; }
    // scale / bias of this tile's couts, 4 per thread, fetched now so that the epilogue never waits on global memory
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }

    const int nsteps = nA + nB;
    // fragment read offsets: row (lane & 31), logical chunk 2*s + (lane >> 5), swizzled with (row >> 2) & 3
    const int fswz = ((lane & 31) >> 2) & 3;
    const int fo0 = (lane & 31) * 64 + (((lane >> 5)) ^ fswz) * 16;
    const int fo1 = (lane & 31) * 64 + ((2 + (lane >> 5)) ^ fswz) * 16;
    // weight fragments: 128-byte rows, hi chunk 2*s + (lane >> 5), lo chunk = hi chunk + 4, both XOR (row >> 1) & 7
    const int bswz = ((lane & 31) >> 1) & 7;
    const int fb0 = (lane & 31) * 128 + (((lane >> 5)) ^ bswz) * 16;
    const int fb1 = (lane & 31) * 128 + ((2 + (lane >> 5)) ^ bswz) * 16;
    const unsigned char* fa = lds + wave * 32 * 64;
    half8_t h_a0, h_a1, h_b0, h_b1;

// One K-step of MFMAs.  All fragment reads of the step are written first and the MFMAs after them; the
// sched_group_barrier sequence then tells the scheduler to emit them interleaved (first operand pair, then
// "next pair of reads + 3 MFMAs" repeatedly), so that LDS latency is covered by the wave's own MFMAs instead
// of four exposed lgkmcnt(0) waits per step (cdna guide T19).
// the four fragments the step's first MFMA triple needs are requested BEFORE the step's DMA is issued (CF_HEAD): the
// ~100-150 cycles of LDS latency then pass under the ~740 cycles the wave spends issuing its global_load_lds
// instructions (profiles/r02q_conv_kstep_stamps.txt) instead of in front of the first MFMA.
#define CF_HEAD(STAGEOFF)                                                                                 \
    {                                                                                                     \
        h_a0 = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + fo0);                                   \
        h_a1 = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + A_PLANE + fo0);                         \
        h_b0 = *reinterpret_cast<const half8_t*>(lds + (STAGEOFF) + 2 * A_PLANE + fb0);                    \
        h_b1 = *reinterpret_cast<const half8_t*>(lds + (STAGEOFF) + 2 * A_PLANE + (fb0 ^ 64));             \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define CF_COMPUTE(STAGEOFF)                                                                              \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        half8_t fa_[2][2], fb_[2][TN][2];                                                                  \
        _Pragma("unroll") for (int s = 0; s < 2; ++s) {                                                    \
            const int fo = s ? fo1 : fo0;                                                                  \
            if (s == 0) { fa_[0][0] = h_a0; fa_[0][1] = h_a1; }                                            \
            else {                                                                                         \
                fa_[s][0] = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + fo);                       \
                fa_[s][1] = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + A_PLANE + fo);             \
            }                                                                                              \
            _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                               \
                const int fbo = s ? fb1 : fb0;                                                             \
                const unsigned char* bb = lds + (STAGEOFF) + 2 * A_PLANE + t * 32 * 128;                   \
                if (s == 0 && t == 0) { fb_[0][0][0] = h_b0; fb_[0][0][1] = h_b1; }                        \
                else {                                                                                     \
                    fb_[s][t][0] = *reinterpret_cast<const half8_t*>(bb + fbo);                            \
                    fb_[s][t][1] = *reinterpret_cast<const half8_t*>(bb + (fbo ^ 64));                     \
                }                                                                                          \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int s = 0; s < 2; ++s)                                                      \
            _Pragma("unroll") for (int t = 0; t < TN; ++t) {                                               \
                /* weight fragment first: the accumulator is D[cout][pixel] (conv_epilogue.h).  The two    \
                   updates of acc_corr[t] are kept one MFMA apart (TN > 1: by the next tile's main MFMA) so  \
                   that no MFMA waits on the result of the one issued just before it */                    \
                acc_corr[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][1], fa_[s][0], acc_corr[0][t], 0, 0, 0); \
                acc_main[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][0], acc_main[0][t], 0, 0, 0); \
                acc_corr[0][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fb_[s][t][0], fa_[s][1], acc_corr[0][t], 0, 0, 0); \
            }                                                                                              \
        /* a(s0), b(s0, t0) are already in registers (CF_HEAD) */                                          \
        _Pragma("unroll") for (int i = 0; i < TN - 1; ++i) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);         /* b(s0, t i+1) */                  \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);         /* MFMAs (s0, t i) */               \
        }                                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);             /* a(s1), b(s1, t0) */              \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);             /* MFMAs (s0, t TN-1) */            \
        _Pragma("unroll") for (int i = 0; i < TN - 1; ++i) {                                               \
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                             \
            __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                             \
        }                                                                                                  \
        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                 \
    }

    // ablation switches for tools/conv_ablate.py (bits above the segment count; never set by the product path):
    // 0x100 no DMA after the first stage, 0x200 no fragment reads / MFMAs, 0x400 no epilogue
    const bool ab_dma = !(p.nseg & 0x100), ab_mma = !(p.nseg & 0x200);
    // measurement switch 0x800 (tools/conv_stamps.py): workgroup 0 writes s_memtime stamps of its first 96 K-steps to the
    // int buffer passed in seg[3].dst -- per step: loop top | DMA landed | barrier passed | DMA issued | MFMAs issued
    // (a separate template instance, DBG = true: the product kernel carries none of this)
    const bool dbg = DBG && (p.nseg & 0x800) && blockIdx.x == 0;
    int* dbuf = reinterpret_cast<int*>(p.seg[3].dst);
#define CF_STAMP(STEP, K)                                                                                  \
    if (DBG && dbg && lane == 0 && (STEP) < 96) dbuf[wave * 512 + (STEP) * 5 + (K)] = (int)__builtin_readcyclecounter();
    if (dbg && lane == 0) dbuf[wave * 512 + 480] = (int)(__builtin_amdgcn_s_memrealtime());
    CF_TAP()
    CF_ISSUE(0)
    int step = 0;
    for (; step + 1 < nsteps; step += 2) {
        CF_STAMP(step, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CF_STAMP(step, 1)
        __syncthreads();
        CF_STAMP(step, 2)
        CF_HEAD(0)
        if (ab_dma) CF_ISSUE(STAGE)                           // step+1 -> stage 1
        CF_STAMP(step, 3)
        if (ab_mma) CF_COMPUTE(0)
        CF_STAMP(step, 4)
        CF_STAMP(step + 1, 0)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        CF_STAMP(step + 1, 1)
        __syncthreads();
        CF_STAMP(step + 1, 2)
        CF_HEAD(STAGE)
        if (step + 2 < nsteps && ab_dma) CF_ISSUE(0)          // step+2 -> stage 0
        CF_STAMP(step + 1, 3)
        if (ab_mma) CF_COMPUTE(STAGE)
        CF_STAMP(step + 1, 4)
    }
    if (dbg && lane == 0) dbuf[wave * 512 + 481] = (int)(__builtin_amdgcn_s_memrealtime());
    if (step < nsteps) {                                      // odd tail: its data sits in stage 0
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        CF_HEAD(0)
        CF_COMPUTE(0)
    }
    __syncthreads();
    if ((p.nseg & 0x400) && p.M > 0) return;                  // (M > 0 always: keeps the accumulators live)
    constexpr int ETW = TN > 1 ? 2 : 1;                       // accumulator tiles staged together: 128-byte runs
    constexpr int EPI0 = 4 * conv_epi::Staging<ETW>::BYTES;   // scale / bias / chunk descriptors behind the staging tiles
    static_assert(EPI0 + conv_epi::EpiArea<BN>::BYTES <= LDS_BYTES, "epilogue staging must fit the operand LDS");
    conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre);
    __syncthreads();
    conv_epi::store_tiles_desc<TN, ETW>(p, acc_main, acc_corr, lds + wave * conv_epi::Staging<ETW>::BYTES, lds + EPI0, m0 + wave * 32);
}


int tise_conv_pipe_launch(const tise_conv_args* a, int cfg, void* stream);   // conv_pipe.hip

extern "C" int tise_conv_split_f16(const ConvArgs* args, int tn, void* stream) {
    if (!args || !args->x || !args->w || !args->scale || !args->bias || (args->nseg & 0xff) < 1 || (args->nseg & 0xff) > 4 ||
        args->Cin % 16 != 0 || args->Cin < 32 || (args->Kpad % CS_BK != 0 && !(tn & 512)) || args->M <= 0)
        return TISE_ERR_INVALID_ARG;
    // the epilogues work on 8-cout chunks and 16-byte stores: segments must start on multiples of 8 couts and land
    // on 16-byte boundaries (fp16 planes: 8 elements, fp32: 4 elements)
    for (int i = 0; i < (args->nseg & 0xff); ++i) {
        const tise_conv_seg& g = args->seg[i];
        const int al = g.mode == 0 ? 8 : 4;
        if (!g.dst || g.c0 % 8 != 0 || g.c1 <= g.c0 || g.off % al != 0 || g.ld % al != 0 || (g.mode == 0 && g.plane % 8 != 0) ||
            (reinterpret_cast<uintptr_t>(g.dst) & 15) != 0 || (i > 0 && g.c0 != args->seg[i - 1].c1) || (g.mode != 0 && g.mode != 1))
            return TISE_ERR_INVALID_ARG;
    }
    if (args->seg[0].c0 != 0) return TISE_ERR_INVALID_ARG;
    if (tn & 512) return tise_conv_pipe_launch(args, tn & 255, stream);   // persistent 3-stage kernel, weights [tap][Cin_pad]
    const bool glds = (tn & (16 | 128)) != 0;
    // fast path: K order (tap, full 32-channel block) then paired 16-channel tails (see the kernel); Kpad says which
    const int fast_kpad = (args->KH * args->KW * (args->Cin / 32) + ((args->Cin & 16) ? (args->KH * args->KW + 1) / 2 : 0)) * 32;
    const bool fast = (tn & 128) != 0 && args->Cin % 16 == 0 && args->Kpad == fast_kpad && args->M < 0x7fffff00LL;
    tn &= 15;
    const int bn = 32 * tn;
    const int bm = CS_BM;
    const long long tiles = ((args->M + bm - 1) / bm) * ((args->Cout + bn - 1) / bn);
    if (tiles > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)tiles), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (fast && (args->nseg & 0x800)) {                       // tools/conv_stamps.py: the instrumented instance
        switch (tn) {
            case 3: hipLaunchKernelGGL((conv_split_fast_kernel<3, true>), grid, block, 0, st, *args); break;
            case 4: hipLaunchKernelGGL((conv_split_fast_kernel<4, true>), grid, block, 0, st, *args); break;
            default: return TISE_ERR_INVALID_ARG;
        }
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    if (fast) {
        switch (tn) {
            case 1: hipLaunchKernelGGL(conv_split_fast_kernel<1>, grid, block, 0, st, *args); break;
            case 2: hipLaunchKernelGGL(conv_split_fast_kernel<2>, grid, block, 0, st, *args); break;
            case 3: hipLaunchKernelGGL(conv_split_fast_kernel<3>, grid, block, 0, st, *args); break;
            case 4: hipLaunchKernelGGL(conv_split_fast_kernel<4>, grid, block, 0, st, *args); break;
            case 5: hipLaunchKernelGGL(conv_split_fast_kernel<5>, grid, block, 0, st, *args); break;
            default: return TISE_ERR_INVALID_ARG;
        }
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    if (glds) {
        switch (tn) {
            case 1: hipLaunchKernelGGL(conv_split_glds_kernel<1>, grid, block, 0, st, *args); break;
            case 2: hipLaunchKernelGGL(conv_split_glds_kernel<2>, grid, block, 0, st, *args); break;
            case 3: hipLaunchKernelGGL(conv_split_glds_kernel<3>, grid, block, 0, st, *args); break;
            case 4: hipLaunchKernelGGL(conv_split_glds_kernel<4>, grid, block, 0, st, *args); break;
            case 5: hipLaunchKernelGGL(conv_split_glds_kernel<5>, grid, block, 0, st, *args); break;
            default: return TISE_ERR_INVALID_ARG;
        }
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    switch (tn) {
        case 2: hipLaunchKernelGGL(conv_split_kernel<2>, grid, block, 0, st, *args); break;
        case 3: hipLaunchKernelGGL(conv_split_kernel<3>, grid, block, 0, st, *args); break;
        case 4: hipLaunchKernelGGL(conv_split_kernel<4>, grid, block, 0, st, *args); break;
        case 5: hipLaunchKernelGGL(conv_split_kernel<5>, grid, block, 0, st, *args); break;
        default: return TISE_ERR_INVALID_ARG;
    }
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

TISE_DEFINE_SPLIT_FLAG_READER(tise_internal_split_flag_conv_split)
