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
// Activations live in HBM in that split form -- the same 4 bytes per element as fp32, NHWC, the two halves of
// every 32-channel block side by side (common.h: tise_ilv_off) -- so the hot loop loads fp16 fragments directly and
// never converts.  The epilogue applies the folded-BatchNorm scale/bias and ReLU, re-splits the fp32 accumulators
// and writes straight into a channel slice of the consumer's tensor (block concat buffer / next conv input), or
// writes raw fp32 for the pool branch (whose 3x3 average runs after the 1x1 conv, trunk_ops.hip).
//
// GEMM view: M = N*OH*OW output pixels, N = Cout, K = KH*KW*Cin.  Workgroup = 256 threads = 4 waves stacked along
// M; tile = 128 pixels x (32*TN) couts x 32 k per step; each wave owns 32 x 32*TN: TN accumulator pairs (main,
// corr) of v_mfma_f32_32x32x16_f16.  Three kernels: the generic one recomputes every source address per K-step (the
// reference of the tests, any M); the default one hoists the addressing out of the K loop; the row-window one
// fetches the pixel operand once for all taps of a filter row (stride-1 layers with KW > 1).  conv_pipe.hip holds a
// fourth for Conv2d_2a.  (Round 1's register-staged, 3-stage, window and wave-specialised variants and round 2's
// big-tile / interleaved-issue experiments tied or lost and were removed; DESIGN.md section 4a, profiles/.)
#include <hip/hip_fp16.h>
#include "common.h"
#include "conv_epilogue.h"

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));   // native vector: stays in registers (SROA)

#define CS_BM 128
#define CS_BK 32

typedef tise_conv_seg ConvSeg;
typedef tise_conv_args ConvArgs;

// ------------------------------------------------------------------------------------------------
// Generic kernel: operands go global -> LDS directly (global_load_lds_dwordx4, no staging registers), two LDS
// stages, one barrier per K-step: the loads of step s+1 are issued right after the barrier of step s and
// stay in flight under that step's MFMAs.  K = (kh, kw, cin) with cin fastest, 32 per step (a step may straddle
// two taps at 16-channel granularity); weights [plane][Cout_pad][K_pad].  In LDS the hi and lo halves are separate
// planes of unpadded 64-byte rows (a DMA wave-instruction writes 1 KiB = 16 rows linearly); bank conflicts of the
// ds_read_b128 fragment reads are removed by an XOR swizzle of the 16-byte chunk index with (row >> 2) & 3,
// applied on the SOURCE address of the DMA and on the read address (cdna guide rule 21).  Out-of-image taps and
// tails read a zero page instead.
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
        ibase[jj] = (long long)n * p.H * p.W * p.Cin * 2;     // a pixel is 2 * Cin fp16 elements
    }
    int a_c = unit * 16, a_kh = 0, a_kw = 0, a_k = unit * 16;
    while (a_c >= p.Cin) { a_c -= p.Cin; if (++a_kw == p.KW) { a_kw = 0; ++a_kh; } }

#define CG_ISSUE(STEP, STAGEBASE)                                                                         \
    {                                                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) {                                                 \
            const int ih = ih0[jj] + a_kh, iw = iw0[jj] + a_kw;                                            \
            const bool ok = rok[jj] && a_k < p.K && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;            \
            const _Float16* sh = xg + ibase[jj] + ((long long)ih * p.W + iw) * (2 * p.Cin) + tise_ilv_off(a_c + sub8, p.Cin); \
            const _Float16* sl = sh + tise_ilv_second(a_c, p.Cin);                                         \
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

    conv_epi::Acc16 acc_main[1][TN], acc_corr[1][TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { conv_epi::acc_zero(acc_main[0][t]); conv_epi::acc_zero(acc_corr[0][t]); }
    // scale / bias of this tile's couts, 4 per thread, fetched now so that the epilogue never waits on global memory
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }

    const int nsteps = p.Kpad / CS_BK;
    // fragment read (v_mfma_f32_16x16x32_f16: 16 rows x 32 K): row (lane & 15) of a 16-row half, logical chunk (lane >> 4),
    // swizzled by (row >> 2) & 3 (unchanged by the +16 rows of the second half)
    const int frow = (lane & 15) * 64;
    const int choff = (((lane >> 4)) ^ (((lane & 15) >> 2) & 3)) * 16;
    CG_ISSUE(0, lds)
    for (int step = 0; step < nsteps; ++step) {
        unsigned char* cur = lds + (step & 1) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // my DMA pieces of this stage have landed
        __syncthreads();                                       // everyone's have; everyone left the other stage
        if (step + 1 < nsteps) CG_ISSUE(step + 1, lds + ((step + 1) & 1) * STAGE)
        half8_t a_[2][2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            const unsigned char* ap = cur + (wave * 32 + pi * 16) * 64 + frow + choff;
            a_[pi][0] = *reinterpret_cast<const half8_t*>(ap);
            a_[pi][1] = *reinterpret_cast<const half8_t*>(ap + A_PLANE);
        }
#pragma unroll
        for (int g = 0; g < 2 * TN; ++g) {
            const unsigned char* bp = cur + 2 * A_PLANE + g * 16 * 64 + frow + choff;
            const half8_t b_hi = *reinterpret_cast<const half8_t*>(bp);
            const half8_t b_lo = *reinterpret_cast<const half8_t*>(bp + B_PLANE);
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                // weight fragment first: the accumulator is D[cout][pixel] (conv_epilogue.h)
                float4_t& cm = acc_main[0][g >> 1].v[g & 1][pi];
                float4_t& cc = acc_corr[0][g >> 1].v[g & 1][pi];
                cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_hi, a_[pi][0], cm, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_lo, a_[pi][0], cc, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_hi, a_[pi][1], cc, 0, 0, 0);
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
// Default kernel ("fast"): the generic kernel with the address arithmetic taken out of the K loop and 128-byte
// operand rows.  rocprofv3 counters on the generic kernel (SQ_INSTS_VALU / SQ_VALU_MFMA_BUSY_CYCLES) showed ~130
// vector and ~55 scalar instructions per wave per K-step next to 24 MFMAs: the per-step recomputation of every DMA
// source address (64-bit multiplies, tap decode, bounds checks) cost about as many issue cycles as the MFMAs.
// The K loop is (tap, 32-channel block): a K-step is ONE 128-byte line [hi x32 | lo x32] per pixel and per cout
// (activation layout: common.h; weights packed [cout][k / 32][hi 32 | lo 32], conv_split.py), source pointers are
// rebuilt only when the tap changes (wave-uniform branch) and otherwise advance by 128 bytes; the two LDS stages
// are addressed with compile-time offsets (loop unrolled by two).  The LDS-DMA path moves 128-byte rows at
// 38 B/clk/CU against 30 for 64-byte rows (profiles/r01g_glds_rate_microbench.txt).  LDS rows are 128 B too; the
// ds_read_b128 fragments stay conflict-free with the chunk index XOR-ed with tise_lds_swz(row) = ((row >> 1) & 3) << 1
// (common.h): a b128 group of 16 lanes reads rows r0 .. r0+3, r0+12 .. r0+15 at K group g and r0+4 .. r0+11 at g ^ 1 (the
// 16-row x 32-K fragment of v_mfma_f32_16x16x32_f16); the four rows of one parity within 8 rows get four different chunk
// pairs, and the row 8 further has the other K group: 16 distinct 16-byte bank groups for EVERY r0 (the window kernels
// read at arbitrary tap offsets; round 2's (row >> 1) & 7, chosen for the 32-row fragments of 32x32x16, costs this
// shape a 2-way conflict whenever r0 is not a multiple of 4: 19 % of all indexed-LDS cycles, profiles/r03l).
// CBT = true (round 4): K order (32-channel block, tap) instead of (tap, block) for UNPADDED multi-tap layers with
// Cin % 32 == 0 (the stride-2 3x3 layers of Mixed_6a / 7a).  In tap-major order a pixel's 128-byte line of one channel
// block is wanted again by the next tap that covers the pixel nine K-steps later -- 9 x 64 workgroups x 16 KB per XCD is more
// than its 4 MB of L2, so the line comes over the fabric again (3.3 x the algorithmic reads on 35 x 35 x 288 -> 384 s2,
// profiles/r04h_conv_hbm_traffic_by_layer.txt); in block-major order the neighbouring taps follow one K-step apart.  No
// tap of a valid output row can leave the image, so a tap change is ONE wave-uniform byte offset added to the four
// always-loadable pointers (rows beyond M stay on the zero page).  Same products, another summation order: results
// differ from the tap-major kernel in the last bits (as the row-window kernel's do).
// (Round 5's two other K-loop forms of this kernel -- DUO: two tiles per 512-thread workgroup in enforced anti-phase, -0.3 %;
// EARLY: counted-wait K-step with two bare barriers, +0.3 %, inside the noise -- are kept as tools/probes/conv_duo_early.patch,
// DESIGN.md section 4e; round 6 took them out of the library.)
template <int TN, bool DBG = false, bool CBT = false>
__global__ __launch_bounds__(256, 2) void conv_split_fast_kernel(const ConvArgs p) {
    constexpr int BN = 32 * TN;
    constexpr int A_BYTES = CS_BM * 128, B_BYTES = BN * 128;   // a stage: 128 pixel rows + BN cout rows of 128 B
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int T_BYTES = CS_BM * (BN * 2 + 16);
    constexpr int LDS_BYTES = (2 * STAGE > T_BYTES) ? 2 * STAGE : T_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // scalar: LDS / DMA addresses stay on the SALU
    // 32-bit index arithmetic throughout the prologue (the launcher sends M >= 2^31 to the generic kernel): the 64-bit
    // divisions of the generic kernel cost several hundred instructions per workgroup, ~15 % of a 27-step tile
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

    // A DMA piece = 8 rows x 128 B: lane i fills row (i >> 3), physical 16-byte chunk (i & 7) and fetches the logical
    // chunk (i & 7) ^ tise_lds_swz(row), chunks 0-3 = hi, 4-7 = lo.  A piece starts on a multiple of 8 rows, so
    // tise_lds_swz(row) = (lane >> 4) << 1 for every piece: one chunk value per lane.
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_conv_zero_page);
    const int pix_bytes = p.Cin * 4;
    const unsigned char* pbase[4];                            // (n, oh*SH - PH, ow*SW - PW) of the piece's pixel: tap (0, 0)
    int ihw[4];                                               // its (ih0, iw0), packed 16 + 16 bits, biased by 0x4000
    {
        const unsigned ohw = (unsigned)(p.OH * p.OW), M32 = (unsigned)p.M;
        const unsigned pix0 = tile_m * CS_BM + wave * 32 + (lane >> 3);
        const unsigned pp = pix0 < M32 ? pix0 : 0u;
        unsigned n = pp / ohw;
        const unsigned rem = pp - n * ohw;
        unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int ih0 = (int)oh * p.SH - p.PH, iw0 = (int)ow * p.SW - p.PW;
            const bool rok = pix0 + 8 * jj < M32;
            // a row beyond M gets ih0 = -0x4000: no tap of it is ever inside the image
            ihw[jj] = rok ? ((ih0 + 0x4000) << 16) | (iw0 + 0x4000) : 0;
            pbase[jj] = reinterpret_cast<const unsigned char*>(p.x) +
                        (((long long)n * p.H + ih0) * p.W + iw0) * pix_bytes;
            ow += 8;                                          // next piece: 8 pixels further
            while (ow >= (unsigned)p.OW) { ow -= (unsigned)p.OW; if (++oh == (unsigned)p.OH) { oh = 0; ++n; } }
        }
    }
    // weight sources of this wave's TN DMA pieces (8 couts x 128 B each): one scalar base + a 32-bit byte offset
    // per lane and piece (the weights of a layer are far below 4 GB), advanced by 128 per K-step
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (long long)n0 * p.Kpad * 4;
    unsigned pb[TN];
    int pb_off[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int q = wave * TN + i;                                   // 8-cout block of the tile
        const int r = q * 8 + (lane >> 3);                             // cout row within the tile
        const int c = (lane & 7) ^ tise_lds_swz(r);                    // logical chunk: 0-3 hi, 4-7 lo
        pb[i] = (unsigned)r * (unsigned)p.Kpad * 4u + c * 16;
        pb_off[i] = A_BYTES + q * 1024;
    }
    // pixel-operand pointers for the current tap: ALWAYS loadable (the zero page when the tap falls outside the
    // image or the row outside M) with a per-lane advance of 128 bytes or 0, so that a K-step issues its DMA
    // straight from the registers and spends one 64-bit add per pointer (the null-pointer selects this replaces
    // were 16 of the ~23 vector instructions per step; SQ_INSTS_VALU / SQ_INSTS_MFMA was 3.4-5.7)
    const unsigned char* pa[4];
    int pa_inc[4];                                            // bytes
    // K order.  Phase A: (tap, full 32-channel block), nA = KH*KW*(Cin/32) steps.  Phase B (Cin % 32 == 16 only):
    // the 16-channel tails of the taps, TWO TAPS PER STEP -- chunks 0,1 (hi) and 4,5 (lo) of a step carry the tail of
    // tap 2j, chunks 2,3 and 6,7 that of tap 2j+1 (lanes pick their tap by their chunk) -- so no MFMA runs on padding
    // except in the last step of an odd tap count.  The weights are packed in the same order (conv_split.py).
    int kh = 0, kw = 0, cblk = 0, istep = 0;
    const int ncblk = p.Cin / CS_BK, ntaps = p.KH * p.KW;
    const int nA = ntaps * ncblk;
    const int nB = (p.Cin & 16) ? (ntaps + 1) / 2 : 0;
    const int cl0 = (lane & 7) ^ ((lane >> 4) << 1), cl1 = cl0;   // logical chunk of this lane in every piece
// pointers of piece JJ for tap (KH_, KW_), CHB_ bytes into the pixel
#define CF_TAP_ONE(JJ, KH_, KW_, CHB_, VALID_)                                                            \
    {                                                                                                     \
        const int ih = (ihw[JJ] >> 16) - 0x4000 + (KH_), iw = (ihw[JJ] & 0xffff) - 0x4000 + (KW_);         \
        const bool ok = (VALID_) && ih >= 0 && ih < p.H && iw >= 0 && iw < p.W;                            \
        const unsigned char* src = pbase[JJ] + ((KH_) * p.W + (KW_)) * pix_bytes + (CHB_);                 \
        pa[JJ] = ok ? src : zp;                                                                            \
        pa_inc[JJ] = ok ? 128 : 0;                                                                         \
    }
#define CF_TAP()                                                                                          \
    _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) CF_TAP_ONE(jj, kh, kw, ((jj & 1) ? cl1 : cl0) * 16, kh < p.KH)
// phase B step j: a lane with logical chunk c serves tap 2j + ((c >> 1) & 1); its 16 bytes are the (c & 1) half of
// the hi (c < 4) or lo 16-channel run of the tail block, which starts 128 * ncblk bytes into the pixel
#define CF_TAP_B(J)                                                                                       \
    _Pragma("unroll") for (int par = 0; par < 2; ++par) {                                                  \
        const int c_ = par ? cl1 : cl0;                                                                    \
        const int tl = 2 * (J) + ((c_ >> 1) & 1);                                                          \
        const int lkh = tl / p.KW, lkw = tl - lkh * p.KW;                                                  \
        const int chb = ncblk * 128 + (c_ >> 2) * 32 + (c_ & 1) * 16;                                      \
        CF_TAP_ONE(par, lkh, lkw, chb, tl < ntaps)                                                         \
        CF_TAP_ONE(par + 2, lkh, lkw, chb, tl < ntaps)                                                     \
    }
#define CF_ISSUE(STAGEOFF)                                                                                \
    {                                                                                                     \
        _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) {                                                 \
            /* a local on purpose: with array elements as direct builtin arguments hipcc (ROCm 7.2) silently \
               drops the host-side launch stub of this template */                                         \
            const unsigned char* sa = pa[jj];                                                              \
            /* measurement switch 0x1000 (instrumented instance only): the pixel operand is fetched every 7th step */ \
            if (!(DBG && (p.nseg & 0x1000)) || istep % 7 == 0)                                             \
                __builtin_amdgcn_global_load_lds(sa, (lds_ptr_t)(lds + (STAGEOFF) + (4 * wave + jj) * 1024), 16, 0, 0); \
            pa[jj] = sa + pa_inc[jj];                                                                      \
        }                                                                                                  \
        _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                   \
            const unsigned char* sw_ = wbase + pb[i];                                                      \
            unsigned char* dw_ = lds + (STAGEOFF) + pb_off[i];                                             \
            __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)dw_, 16, 0, 0);                               \
            pb[i] += 128;                                    /* next K-step: 128 bytes further */          \
        }                                                                                                  \
        ++istep;                                             /* the step whose pointers are prepared now */ \
        if constexpr (CBT) {                                 /* (block, tap) order: scalar offset to the next tap's pixel */ \
            int delta = pix_bytes - 128;                     /* next tap in the filter row (the issue above advanced 128 already) */ \
            if (++kw == p.KW) {                                                                            \
                kw = 0;                                                                                    \
                delta = (p.W - p.KW + 1) * pix_bytes - 128;  /* first tap of the next filter row */        \
                if (++kh == p.KH) { kh = 0; delta = -((p.KH - 1) * p.W + (p.KW - 1)) * pix_bytes; }   /* tap (0, 0) of the next block: + 128 */ \
            }                                                                                              \
            _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) pa[jj] += (long long)(pa_inc[jj] ? delta : 0); \
        } else if (istep >= nA) {                            /* wave-uniform */                            \
            if (istep < nA + nB) CF_TAP_B(istep - nA)                                                      \
        } else if (++cblk == ncblk) {                        /* next filter tap */                         \
            cblk = 0;                                                                                      \
            if (++kw == p.KW) { kw = 0; ++kh; }                                                            \
            CF_TAP()                                                                                       \
        }                                                                                                  \
    }

    // wave layout.  TN = 4: 2 x 2 -- a wave owns 64 pixels x 64 couts (16 fragment reads per K-step instead of the 20 of
    // 32 pixels x 128 couts; the probe puts 6-7 % between the two at the power limit, profiles/r03k_mfma_shape_probe.txt);
    // other widths: 4 x 1 -- 32 pixels x all TN * 32 couts
    constexpr bool W22 = TN == 4;
    constexpr int NPH = W22 ? 4 : 2;                          // 16-pixel fragment halves per wave
    constexpr int NCH = W22 ? TN : 2 * TN;                    // 16-cout fragment halves per wave
    constexpr int RG = NPH / 2, TNW = NCH / 2;                // as 32 x 32 tiles: row groups x cout tiles
    const int wm = W22 ? (wave & 1) : wave, wn = W22 ? (wave >> 1) : 0;
    conv_epi::Acc16 acc_main[RG][1][TNW], acc_corr[RG][1][TNW];
#pragma unroll
    for (int r = 0; r < RG; ++r)
#pragma unroll
        for (int t = 0; t < TNW; ++t) { conv_epi::acc_zero(acc_main[r][0][t]); conv_epi::acc_zero(acc_corr[r][0][t]); }
    // scale / bias of this tile's couts, 4 per thread, fetched now so that the epilogue never waits on global memory
    // (TN = 5 has no eight registers to hold them through the K loop: it fetches them afterwards)
    constexpr bool PRE = TN < 5;
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (PRE && tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }

    const int nsteps = nA + nB;
    // fragments of both operands (v_mfma_f32_16x16x32_f16: a fragment is 16 rows x 32 K): 128-byte rows, row (lane & 15) of
    // a 16-row half, hi chunk (lane >> 4), lo chunk = hi chunk + 4, both XOR tise_lds_swz(row); the second half of a 32-row
    // tile is 2048 bytes further (16 rows leave the swizzle unchanged), the next tile 4096
    const int f16o = (lane & 15) * 128 + (((lane >> 4) ^ tise_lds_swz(lane & 15)) << 4);
    const unsigned char* fa = lds + wm * NPH * 2048;
    const int fbw = A_BYTES + wn * NCH * 2048;                // this wave's first cout half in the stage
    half8_t h_a[NPH][2], h_b[2];                              // a: [pixel half][hi / lo], b: (cout half 0)[hi / lo]

// One K-step of MFMAs: NCH groups (one 16-cout half each) of 3 NPH -- per pixel half corr(b_lo, a_hi), main(b_hi, a_hi),
// corr(b_hi, a_lo) -- each preceded in issue order by the two fragment reads of the NEXT group (sched_group_barrier), so
// the LDS latency of a weight fragment passes under the MFMAs (96 / 192 cycles) before its own.
// The fragments the step's first group needs are requested BEFORE the step's DMA is issued (CF_HEAD): the
// ~100-150 cycles of LDS latency then pass under the ~740 cycles the wave spends issuing its global_load_lds
// instructions (profiles/r02q_conv_kstep_stamps.txt) instead of in front of the first MFMA.
#define CF_HEAD(STAGEOFF)                                                                                 \
    {                                                                                                     \
        _Pragma("unroll") for (int pi = 0; pi < NPH; ++pi) {                                               \
            h_a[pi][0] = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + pi * 2048 + f16o);            \
            h_a[pi][1] = *reinterpret_cast<const half8_t*>(fa + (STAGEOFF) + pi * 2048 + (f16o ^ 64));     \
        }                                                                                                  \
        h_b[0] = *reinterpret_cast<const half8_t*>(lds + (STAGEOFF) + fbw + f16o);                         \
        h_b[1] = *reinterpret_cast<const half8_t*>(lds + (STAGEOFF) + fbw + (f16o ^ 64));                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
    }
#define CF_COMPUTE(STAGEOFF)                                                                              \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        half8_t fb_[NCH][2];                                                                               \
        _Pragma("unroll") for (int g = 0; g < NCH; ++g) {                                                  \
            if (g == 0) { fb_[0][0] = h_b[0]; fb_[0][1] = h_b[1]; }                                        \
            else {                                                                                         \
                const unsigned char* bb = lds + (STAGEOFF) + fbw + g * 2048;                               \
                fb_[g][0] = *reinterpret_cast<const half8_t*>(bb + f16o);                                  \
                fb_[g][1] = *reinterpret_cast<const half8_t*>(bb + (f16o ^ 64));                           \
            }                                                                                              \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < NCH; ++g)                                                    \
            _Pragma("unroll") for (int pi = 0; pi < NPH; ++pi) {                                           \
                /* weight fragment first: the accumulator is D[cout][pixel] (conv_epilogue.h) */           \
                float4_t& cm = acc_main[pi >> 1][0][g >> 1].v[g & 1][pi & 1];                              \
                float4_t& cc = acc_corr[pi >> 1][0][g >> 1].v[g & 1][pi & 1];                              \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][1], h_a[pi][0], cc, 0, 0, 0);           \
                cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][0], h_a[pi][0], cm, 0, 0, 0);           \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][0], h_a[pi][1], cc, 0, 0, 0);           \
            }                                                                                              \
        _Pragma("unroll") for (int g = 0; g < NCH; ++g) {                                                  \
            if (g + 1 < NCH) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);        /* b(g + 1) */      \
            __builtin_amdgcn_sched_group_barrier(0x008, 3 * NPH, 0);                   /* MFMAs (g) */     \
        }                                                                                                  \
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
#define CF_BAR() { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_sched_barrier(0); }
    int step = 0;
    CF_TAP()
    CF_ISSUE(0)
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
    if (!PRE && tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }
    conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre, tid);
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RG; ++r)
        conv_epi::store_tiles_desc<TNW, ETW, false, BN>(p, acc_main[r], acc_corr[r], lds + wave * conv_epi::Staging<ETW>::BYTES, lds + EPI0,
                                                        m0 + wm * NPH * 16 + r * 32, wn * NCH * 2);
}


// ------------------------------------------------------------------------------------------------
// Row-window kernel ("rowwin"): the default kernel for stride-1 layers whose filter is wider than one pixel (the 1x7,
// 1x3, 3x3 and 5x5 layers, padded or not).
// In the default kernel the KW taps of a filter row fetch the SAME input pixels KW times, each time shifted by one
// pixel.  Here the K loop runs (kh, 32-channel block, kw) and the pixel operand of a (kh, block) GROUP is fetched
// once, as a window in "padded-x" coordinates: output pixel i of the tile (row ordinal j = (x0 + i) / OW inside the
// tile) sits at window row i + j * (KW - 1), its tap kw at window row i + j * (KW - 1) + kw; a row ordinal owns
// OW + KW - 1 = W + 2 PW window rows: PW of left padding, the W input pixels, PW of right padding (zero page), and
// rows whose input line oy + kh - PH lies outside the image come from the zero page as well.  No masking in
// registers; the fragment address moves by one row per tap.  The window of the next group is fetched piece by piece
// during the steps of the current one (two window buffers); the weights stream as before (two stages, one K-step
// ahead).  Pixel-operand DMA per group: 128 + J * (KW - 1) rows (J <= 128 / OW + 2 image rows) instead of 128 * KW.
// Cin = 32 n + 16: each kh ends with a TAIL group of the last 16 channels whose steps pair two taps -- K-slice 0
// reads tap 2p at the window row of tap 2p, K-slice 1 tap 2p + 1 one row further, both from the 16-channel columns
// (chunks 0-1 hi, 4-5 lo of the row; the rest of a tail row is fetched from the zero page) -- so the tail costs
// ceil(KW / 2) steps per kh, like the default kernel's paired tails.
// K order differs from the default kernel's (tap-major), so results differ from it in the last bits (fp32 summation
// order); against fp64 both have the same error.
//
// POOLH = true (round 4; Conv2d_4a -> MaxPool2d(3, 2), inception.py:69-70): the HORIZONTAL half of the max-pool is taken in
// this kernel's epilogue -- h(y, ox) = max(f(y, 2 ox), f(y, 2 ox + 1), f(y, 2 ox + 2)) -- and only the (OH, (OW - 3) / 2 + 1)
// map is written (half the bytes); the consumer's pooled-input kernel then takes the three VERTICAL taps instead of nine.
// The three columns of a window are neighbouring tile rows, so the tile is converted into a workgroup-wide fp32 image in
// the operand LDS (f = hi + lo * 2^-11 of the re-split result: what a pool kernel would read back) and a second pass reads
// three rows per window.  Tiles advance by 126 pixels and overlap by two, so that every window lies inside the tile that
// holds its FIRST column (1.6 % of the MFMA work is computed twice).  max is monotone and re-splitting a value that is
// already (hi, lo)-representable returns the same merged value: pooling in two halves gives the bits of pooling at once.
template <int TN, int NP, bool POOLH = false>
__global__ __launch_bounds__(256, 2) void conv_split_rowwin_kernel(const ConvArgs p, const unsigned inv_wp) {
    constexpr int BN = 32 * TN;
    constexpr int MSTEP = POOLH ? 126 : CS_BM;                 // pixels between the first pixels of consecutive tiles
    constexpr int A_BYTES = NP * 4 * 1024;                    // window buffer: NP pieces (8 rows x 128 B) per wave
    constexpr int B_BYTES = BN * 128;
    constexpr int B0 = 2 * A_BYTES;
    constexpr int LDS_BYTES = 2 * A_BYTES + 2 * B_BYTES;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned tiles_n = (unsigned)(p.Cout + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned tile_m = bid / tiles_n;
    const int tile_n = (int)(bid - tile_m * tiles_n);
    const long long m0 = (long long)tile_m * MSTEP;
    const int n0 = tile_n * BN;
    const unsigned char* zp = reinterpret_cast<const unsigned char*>(g_conv_zero_page);
    const int pix_bytes = p.Cin * 4;
    const int Wp = p.OW + p.KW - 1;                           // window rows per image row = W + 2 PW
    const unsigned row0 = (unsigned)(tile_m * MSTEP) / (unsigned)p.OW;     // global output line of the tile's first pixel
    const int x0 = (int)(tile_m * MSTEP - row0 * (unsigned)p.OW);
    const unsigned img0 = row0 / (unsigned)p.OH;
    const int oy0 = (int)(row0 - img0 * (unsigned)p.OH);
    const unsigned nrows_out = (unsigned)p.N * (unsigned)p.OH;

    // window pieces of this wave: piece q = wave + 4 t, window row rho = 8 q + (lane >> 3), 16-byte chunk by piece parity
    const unsigned char* pbase[NP];                           // source of (row, kh = 0, block 0, chunk 0)
    int oyp[NP];                                              // oy - PH + 0x4000 of the row's image line, 0 = zero row
    // every piece of a wave has the wave's parity (q = wave + 4 t), so a lane fetches ONE logical chunk in all of them
    const int cch = (lane & 7) ^ ((lane >> 4) << 1);           // tise_lds_swz of a row inside an 8-row piece
    const int full_off = cch * 16;                            // byte offset of that chunk inside a full block's line
    const int tail_off = (cch >> 2) * 32 + (cch & 1) * 16;    // ... inside the 64-byte tail block [hi x16 | lo x16]
    const bool tail_has = (cch & 3) < 2;                      // chunks 2, 3, 6, 7 of a tail row are zeros
#pragma unroll
    for (int t = 0; t < NP; ++t) {
        const int q = wave + 4 * t;
        const int rho = 8 * q + (lane >> 3);
        const unsigned u = (unsigned)(rho + x0);
        const unsigned j = (u * inv_wp) >> 16;                // u / Wp (exact for u < 2048: host check)
        const int xin = (int)(u - j * (unsigned)Wp) - p.PW;
        unsigned n = img0;
        int oy = oy0 + (int)j;
        while (oy >= p.OH) { oy -= p.OH; ++n; }
        const bool okx = xin >= 0 && xin < p.W && row0 + j < nrows_out;
        oyp[t] = okx ? oy - p.PH + 0x4000 : 0;
        pbase[t] = reinterpret_cast<const unsigned char*>(p.x) +
                   (((long long)n * p.H + (oy - p.PH)) * p.W + xin) * pix_bytes;
    }
    // weights: as in the default kernel (one scalar base + a 32-bit offset per piece, 128 bytes per K-step)
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (long long)n0 * p.Kpad * 4;
    unsigned pb[TN];
    int pb_off[TN];
#pragma unroll
    for (int i = 0; i < TN; ++i) {
        const int q = wave * TN + i;
        const int r = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ tise_lds_swz(r);
        pb[i] = (unsigned)r * (unsigned)p.Kpad * 4u + c * 16;
        pb_off[i] = q * 1024;
    }
    // fragment rows (v_mfma_f32_16x16x32_f16: 16 pixels x 32 K per fragment): this lane's pixels
    // i = 32 wave + 16 pi + (lane & 15), pi = 0, 1, sit at window rows i + j(i) * (KW - 1)
    int arow[2];
#pragma unroll
    for (int pi = 0; pi < 2; ++pi) {
        const int i = wave * 32 + pi * 16 + (lane & 15);
        const unsigned inv_w = 65536u / (unsigned)p.OW + 1u;
        const unsigned ji = ((unsigned)(x0 + i) * inv_w) >> 16;          // (x0 + i) / OW, x0 + i < OW + 128
        arow[pi] = i + (int)ji * (p.KW - 1);
    }
    // weight fragments: row (lane & 15) of a 16-cout half, chunk (lane >> 4); next half 2048 bytes further (as the default kernel)
    const int f16o = (lane & 15) * 128 + (((lane >> 4) ^ tise_lds_swz(lane & 15)) << 4);

    conv_epi::Acc16 acc_main[1][TN], acc_corr[1][TN];
#pragma unroll
    for (int t = 0; t < TN; ++t) { conv_epi::acc_zero(acc_main[0][t]); conv_epi::acc_zero(acc_corr[0][t]); }
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }

    const int ncblk = p.Cin / CS_BK;                          // full 32-channel blocks
    const int has_tail = (p.Cin & 16) ? 1 : 0;
    const int gpk = ncblk + has_tail;                         // groups per kh; group index gpk - 1 is the tail when has_tail
    const int tail_steps = (p.KW + 1) / 2;
    const int ngroups = p.KH * gpk;
    const int nsteps = p.KH * (ncblk * p.KW + has_tail * tail_steps);
    int tmodf[NP], tmodt[NP];                                 // the step of a full / tail group at which window piece t of the next group goes out
#pragma unroll
    for (int t = 0; t < NP; ++t) { tmodf[t] = t % p.KW; tmodt[t] = t % tail_steps; }
// window piece T of group (GKH, GCB) into window buffer AOFF
#define RW_A_PIECE(T, GKH, GCB, AOFF)                                                                     \
    {                                                                                                     \
        const int iy = oyp[T] - 0x4000 + (GKH);                                                            \
        const bool gt_ = has_tail && (GCB) == ncblk;            /* the group is a tail group (wave-uniform) */ \
        const bool ok = (unsigned)iy < (unsigned)p.H && (!gt_ || tail_has);                                \
        const unsigned char* src = pbase[T] + ((GKH) * p.W * pix_bytes + (GCB) * 128 + (gt_ ? tail_off : full_off)); \
        src = ok ? src : zp;                                                                               \
        __builtin_amdgcn_global_load_lds(src, (lds_ptr_t)(lds + (AOFF) + (wave + 4 * (T)) * 1024), 16, 0, 0); \
    }
#define RW_B_ISSUE(BOFF)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < TN; ++i) {                                                       \
        const unsigned char* sw_ = wbase + pb[i];                                                          \
        __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)(lds + (BOFF) + pb_off[i]), 16, 0, 0);            \
        pb[i] += 128;                                                                                      \
    }
// one K-step: weights of the next step, a share of the next group's window, then reads + MFMAs
#define RW_STEP(BCUR, BNEXT)                                                                              \
    {                                                                                                     \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                  \
        __syncthreads();                                                                                   \
        /* a lane's K group (lane >> 4) of the step's 32: full group: chunk (lane >> 4) of tap kw's row; tail group: K groups  \
           0, 1 = the 16 channels (chunks 0, 1) of tap 2 kw, K groups 2, 3 = those of tap 2 kw + 1, one row further */          \
        const int rsel = cur_tail ? 2 * kw + (lane >> 5) : kw;                                             \
        const int csel = cur_tail ? ((lane >> 4) & 1) : (lane >> 4);                                       \
        half8_t fa_[2][2], fb_[2 * TN][2];                                                                 \
        _Pragma("unroll") for (int pi = 0; pi < 2; ++pi) {                                                 \
            const int row = arow[pi] + rsel;                                                               \
            const unsigned char* ap = lds + acur + row * 128 + ((csel ^ tise_lds_swz(row)) << 4);          \
            fa_[pi][0] = *reinterpret_cast<const half8_t*>(ap);                                            \
            fa_[pi][1] = *reinterpret_cast<const half8_t*>(lds + acur + row * 128 + (((csel ^ tise_lds_swz(row)) << 4) ^ 64)); \
        }                                                                                                  \
        fb_[0][0] = *reinterpret_cast<const half8_t*>(lds + (BCUR) + f16o);                                \
        fb_[0][1] = *reinterpret_cast<const half8_t*>(lds + (BCUR) + (f16o ^ 64));                         \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        if (step + 1 < nsteps) RW_B_ISSUE(BNEXT)                                                           \
        if (gnext < ngroups) {                                                                             \
            _Pragma("unroll") for (int t = 0; t < NP; ++t)                                                 \
                if ((cur_tail ? tmodt[t] : tmodf[t]) == kw) RW_A_PIECE(t, nkh, ncb, anext)                 \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        _Pragma("unroll") for (int g = 1; g < 2 * TN; ++g) {                                               \
            const unsigned char* bb = lds + (BCUR) + g * 2048;                                             \
            fb_[g][0] = *reinterpret_cast<const half8_t*>(bb + f16o);                                      \
            fb_[g][1] = *reinterpret_cast<const half8_t*>(bb + (f16o ^ 64));                               \
        }                                                                                                  \
        _Pragma("unroll") for (int g = 0; g < 2 * TN; ++g)                                                 \
            _Pragma("unroll") for (int pi = 0; pi < 2; ++pi) {                                             \
                float4_t& cm = acc_main[0][g >> 1].v[g & 1][pi];                                           \
                float4_t& cc = acc_corr[0][g >> 1].v[g & 1][pi];                                           \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][1], fa_[pi][0], cc, 0, 0, 0);           \
                cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][0], fa_[pi][0], cm, 0, 0, 0);           \
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb_[g][0], fa_[pi][1], cc, 0, 0, 0);           \
            }                                                                                              \
        _Pragma("unroll") for (int g = 0; g < 2 * TN; ++g) {                                               \
            if (g + 1 < 2 * TN) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     /* b(g + 1) */      \
            __builtin_amdgcn_sched_group_barrier(0x008, 6, 0);                         /* MFMAs (g) */     \
        }                                                                                                  \
        __builtin_amdgcn_sched_barrier(0);                                                                 \
        ++step;                                                                                            \
        if (++kw == (cur_tail ? tail_steps : p.KW)) {         /* next group */                             \
            kw = 0;                                                                                        \
            const int tsw = acur; acur = anext; anext = tsw;                                               \
            cur_tail = has_tail && ncb == ncblk;              /* the group just fetched becomes the current one */ \
            ++gnext;                                                                                       \
            if (++ncb == gpk) { ncb = 0; ++nkh; }                                                          \
        }                                                                                                  \
    }

    // prologue: the first group's window and the first step's weights
#pragma unroll
    for (int t = 0; t < NP; ++t) RW_A_PIECE(t, 0, 0, 0)
    RW_B_ISSUE(B0)
    int step = 0, kw = 0, acur = 0, anext = A_BYTES;
    int cur_tail = 0;                                         // the first group of a kh is a full one (Cin >= 32)
    int gnext = 1, nkh = gpk > 1 ? 0 : 1, ncb = gpk > 1 ? 1 : 0;       // the group whose window is fetched during the current one
    while (step < nsteps) {
        RW_STEP(B0, B0 + B_BYTES)
        if (step < nsteps) RW_STEP(B0 + B_BYTES, B0)
    }
    __syncthreads();
    if constexpr (POOLH) {
        // ---- horizontal max-pool epilogue (kernel header) ---------------------------------------------------------
        constexpr int PITCH = BN * 4 + 16;                    // fp32 tile image: 128 pixel rows; 16-byte pad => conflict-free 16-lane groups
        constexpr int EPI0 = CS_BM * PITCH;
        static_assert(EPI0 + conv_epi::EpiArea<BN>::BYTES <= LDS_BYTES, "pooled epilogue image must fit the operand LDS");
        conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre);
        __syncthreads();
        const unsigned char* area = lds + EPI0;
        float vmax = 0.f;
        const int l4 = lane >> 4;
        // the image holds the RAW fp32 results r = relu(acc * scale + bias).  F = merge(split(.)) is monotone, so
        // max_i F(r_i) = F(max_i r_i): the windows take the maximum of raw values and the (1/2 as many) results are split,
        // merged and split again -- exactly split(max_i merge(split(r_i))), what pooling the stored split tensor gives
#pragma unroll
        for (int t = 0; t < TN; ++t)
#pragma unroll
            for (int ci = 0; ci < 2; ++ci) {
                const int ch = 32 * t + 16 * ci + 4 * l4;
                const float4_t sc = *reinterpret_cast<const float4_t*>(area + conv_epi::EpiArea<BN>::SCALE + ch * 4);
                const float4_t bs = *reinterpret_cast<const float4_t*>(area + conv_epi::EpiArea<BN>::BIAS + ch * 4);
#pragma unroll
                for (int pi = 0; pi < 2; ++pi) {
                    float4_t r;
#pragma unroll
                    for (int k = 0; k < 4; ++k) {
                        const float v = (acc_main[0][t].v[ci][pi][k] + acc_corr[0][t].v[ci][pi][k] * (1.0f / 2048.0f)) * sc[k];
                        r[k] = fmaxf(v + bs[k], 0.f);
                        vmax = fmaxf(vmax, r[k]);
                    }
                    *reinterpret_cast<float4_t*>(lds + (wave * 32 + pi * 16 + (lane & 15)) * PITCH + ch * 4) = r;
                }
            }
        tise_flag_split_overflow(vmax);
        __syncthreads();
        // windows whose first column is tile pixel i in [0, 126): item = (pixel PAIR q, 8-cout chunk); of the pixels 2 q and
        // 2 q + 1 the one with an even column can start a window (rows are OW pixels long, so the parity changes from row to
        // row; where a pair straddles a row end -- x = OW - 1 even, then x = 0 -- the second pixel is the candidate)
        const int owp = (p.OW - 3) / 2 + 1;
        const unsigned inv_w = 65536u / (unsigned)p.OW + 1u;
        const long long mleft = p.M - m0;                     // pixels of this tile that exist
        // chunk fastest along the lanes: the BN / 8 lanes of a window write its hi and lo runs as whole 64-byte pieces of the
        // destination lines (pixel fastest -- one 16-byte piece per line and lane -- cost the layer 20 %: partial-line
        // streaming stores)
        constexpr int NCH = BN / 8;
        for (int item = tid; item < 64 * NCH; item += 256) {
            const int q = item / NCH, c8 = item - q * NCH;
            int i = 2 * q;
            unsigned ji = ((unsigned)(x0 + i) * inv_w) >> 16;     // image-row ordinal of pixel i inside the tile
            int x = x0 + i - (int)ji * p.OW;
            if ((x & 1) || x == p.OW - 1) {
                ++i; ++x;
                if (x == p.OW) { x = 0; ++ji; }
            }
            if (!(i < MSTEP && i + 2 < mleft && !(x & 1) && x + 2 < p.OW)) continue;
            const conv_epi::ChunkDesc cd = *reinterpret_cast<const conv_epi::ChunkDesc*>(area + conv_epi::EpiArea<BN>::DESC + c8 * 32);
            if (!cd.valid) continue;
            float mx[8];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const unsigned char* qq = lds + (i + j) * PITCH + c8 * 32;
                const float4_t q0 = *reinterpret_cast<const float4_t*>(qq);
                const float4_t q1 = *reinterpret_cast<const float4_t*>(qq + 16);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    mx[k] = j == 0 ? q0[k] : fmaxf(mx[k], q0[k]);
                    mx[4 + k] = j == 0 ? q1[k] : fmaxf(mx[4 + k], q1[k]);
                }
            }
            half8_t ph, pl;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const _Float16 h0 = (_Float16)mx[k];
                const _Float16 l0 = (_Float16)((mx[k] - (float)h0) * 2048.0f);
                const float f = (float)h0 + (float)l0 * (1.0f / 2048.0f);          // what the split tensor of the full result holds
                ph[k] = (_Float16)f;
                pl[k] = (_Float16)((f - (float)ph[k]) * 2048.0f);
            }
            const unsigned long long pix = (unsigned long long)(row0 + ji) * (unsigned)owp + (unsigned)(x >> 1);
            unsigned char* d = reinterpret_cast<unsigned char*>(cd.base) + pix * (unsigned long long)cd.row_stride;
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, ph), conv_epi::global_ptr(d));
            __builtin_nontemporal_store(__builtin_bit_cast(u32x4_t, pl), conv_epi::global_ptr(d + cd.second));
        }
    } else {
        constexpr int ETW = TN > 1 ? 2 : 1;
        constexpr int EPI0 = 4 * conv_epi::Staging<ETW>::BYTES;
        static_assert(EPI0 + conv_epi::EpiArea<BN>::BYTES <= LDS_BYTES, "epilogue staging must fit the operand LDS");
        conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre);
        __syncthreads();
        conv_epi::store_tiles_desc<TN, ETW>(p, acc_main, acc_corr, lds + wave * conv_epi::Staging<ETW>::BYTES, lds + EPI0, m0 + wave * 32);
    }
}

template <int TN, int NP, bool POOLH = false>
static int launch_rowwin(const ConvArgs* args, hipStream_t st) {
    constexpr int LDS = 2 * NP * 4096 + 2 * 32 * TN * 128;
    static_assert(2 * LDS <= 160 * 1024, "two workgroups per CU");
    static std::atomic<unsigned long long> attr_set{0};
    if (tise_first_use_on_this_device(attr_set)) {
        TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_split_rowwin_kernel<TN, NP, POOLH>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
    }
    const int bn = 32 * TN;
    // POOLH: tiles start every 126 pixels; the last one must hold the first column of the last window (pixel M - 3)
    const long long tiles_m = POOLH ? (args->M - 3) / 126 + 1 : (args->M + CS_BM - 1) / CS_BM;
    const long long tiles = tiles_m * ((args->Cout + bn - 1) / bn);
    if (tiles > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    const unsigned wp = (unsigned)(args->OW + args->KW - 1);
    const unsigned inv_wp = 65536u / wp + 1u;
    hipLaunchKernelGGL((conv_split_rowwin_kernel<TN, NP, POOLH>), dim3((unsigned)tiles), dim3(256), LDS, st, *args, inv_wp);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

// window pieces per wave a layer needs: rows 128 + J (KW - 1) with J = (OW + 126) / OW + 1 image rows at most, + 1: the
// second K-slice of a tail step reads one row beyond its tap (times zero weights when KW is odd: it must be DMA'd data)
static int rowwin_np(const ConvArgs* a) {
    const int J = (a->OW + 126) / a->OW + 1;
    const int rows = 128 + J * (a->KW - 1) + 1;
    return (rows + 31) / 32;
}

static int launch_rowwin_any(const ConvArgs* a, int tn, hipStream_t st, bool poolh = false) {
    const int steps_per_kh = (a->Cin / 32) * a->KW + ((a->Cin & 16) ? (a->KW + 1) / 2 : 0);
    if (a->SH != 1 || a->SW != 1 || a->KW < 2 || a->KW > 8 || a->OW != a->W + 2 * a->PW - a->KW + 1 || a->OW < 1 || a->Cin % 16 != 0 ||
        a->Cin < 32 || a->Kpad != a->KH * steps_per_kh * 32 || a->M >= 0x7fffff00LL || a->H >= 0x3f00 || a->PH >= 0x100)
        return TISE_ERR_INVALID_ARG;
    {   // the multiply-shift divisions of the kernel must be exact over their ranges
        const unsigned wp = (unsigned)(a->OW + a->KW - 1), w = (unsigned)a->OW;
        const unsigned iwp = 65536u / wp + 1u, iw = 65536u / w + 1u;
        for (unsigned u = 0; u < 512 + w; ++u)
            if (((u * iwp) >> 16) != u / wp || ((u * iw) >> 16) != u / w) return TISE_ERR_UNSUPPORTED;
    }
    const int np = rowwin_np(a);
    if (poolh) {                                               // horizontal max-pool in the epilogue (Conv2d_4a's instance)
        if (a->OW < 3 || a->M < 3 || (a->nseg & ~0xff)) return TISE_ERR_INVALID_ARG;
        for (int i = 0; i < (a->nseg & 0xff); ++i)
            if (a->seg[i].mode != 0) return TISE_ERR_INVALID_ARG;
        if (tn == 3 && np == 5) return launch_rowwin<3, 5, true>(a, st);
        if (tn == 3 && np == 6) return launch_rowwin<3, 6, true>(a, st);
        return TISE_ERR_UNSUPPORTED;
    }
    if (np == 5) {
        switch (tn) {
            case 2: return launch_rowwin<2, 5>(a, st);
            case 3: return launch_rowwin<3, 5>(a, st);
            case 4: return launch_rowwin<4, 5>(a, st);
            default: return TISE_ERR_INVALID_ARG;
        }
    }
    if (np == 6) {
        switch (tn) {
            case 2: return launch_rowwin<2, 6>(a, st);
            case 3: return launch_rowwin<3, 6>(a, st);
            case 4: return launch_rowwin<4, 6>(a, st);
            default: return TISE_ERR_INVALID_ARG;
        }
    }
    return TISE_ERR_UNSUPPORTED;
}

// ------------------------------------------------------------------------------------------------
// Pooled-input 1x1 convolution ("poolin"): max_pool2d(3, stride 2) FUSED INTO THE OPERAND LOAD of the 1x1 convolution
// that consumes it -- the two stem pools of the trunk (inception.py:61-66: MaxPool -> Conv2d_3b_1x1; :69-71: MaxPool
// -> Mixed_5b, whose four 1x1 convolutions run as one).  As separate kernels the pool wrote its result (1.4 + 0.9 GB per
// 1000 images), the convolution read it back, and both launches were bound by that traffic (profiles/r02z: max-pools
// 2.8 ms, Conv2d_3b 0.68 ms, Mixed_5b's 1x1 0.6 ms per 1000 images).  Here the pixel operand of a K-step is not DMA'd:
// every thread loads the nine taps of two (pooled pixel, 8-channel) pairs -- 64 contiguous bytes per pixel line and
// four lanes -- takes the maximum of the MERGED values v = hi + lo * 2^-11 (exact in fp32), re-splits it and writes the
// (hi, lo) chunks into the LDS rows the MFMA fragments are read from (the default kernel's 128-byte rows and swizzle).
// The arithmetic is that of maxpool3s2_split_kernel followed by conv_split_fast_kernel -- the same re-split values, the
// same K order and MFMA sequence -- so the result is BIT-IDENTICAL to the two-kernel path (tests).  Weights stream by
// LDS-DMA as in the default kernel (two stages); one pixel-operand buffer (the kernel is bound by the pool's input
// read, and two to three workgroups per CU overlap one another's load and MFMA phases).
// Geometry in the argument struct: H, W = the UN-pooled input, OH, OW = the pooled grid = the convolution's output
// grid, KH = KW = 1, M = N * OH * OW.
// Tile = 64 pooled pixels x 64 * TNW couts per 256-thread workgroup, waves 2 (pixels) x 2 (couts): every thread owns ONE
// (pixel, 8-channel) item per K-step -- its 18 loads go out together and the step costs one memory round trip (the
// first version, 128 pixels x 32 TN couts with two items per thread, spent two round trips per step with its loads
// split in two batches by register pressure and re-pooled Mixed_5b's input once per 128-cout tile: 2.4 TB/s of input
// against the stand-alone pool kernel's 4.8, profiles/r03d_trunk_conv_layers.txt).  TNW = 2: 128 couts (Conv2d_3b's
// 80), three workgroups per CU;  TNW = 4: 256 couts (Mixed_5b's 208 in ONE tile: pooled once), two per CU.
// VT = true (round 4): the input is ALREADY pooled horizontally by its producer (row-window kernel, POOLH): three vertical
// taps at stride 2 instead of nine -- H = the producer's rows, W = OW = the pooled columns.
template <int TNW, bool VT = false>
__global__ __launch_bounds__(256, TNW == 2 ? 3 : 2) void conv_poolin_kernel(const ConvArgs p) {
    constexpr int PM = 64;                                    // pooled pixels per tile
    constexpr int BN = 64 * TNW;
    constexpr int A_BYTES = PM * 128, B_BYTES = BN * 128;
    constexpr int OPER = A_BYTES + 2 * B_BYTES;
    constexpr int ETW = 2;
    constexpr int EPI0 = 4 * conv_epi::Staging<ETW>::BYTES;
    constexpr int EPI_BYTES = EPI0 + conv_epi::EpiArea<BN>::BYTES;
    constexpr int LDS_BYTES = OPER > EPI_BYTES ? OPER : EPI_BYTES;
    __shared__ __attribute__((aligned(1024))) unsigned char lds[LDS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave & 1, wn = wave >> 1;
    const unsigned tiles_n = (unsigned)(p.Cout + BN - 1) / BN;
    const unsigned nwg = gridDim.x;
    unsigned bid = blockIdx.x;
    {
        const unsigned q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = ((xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const unsigned tile_m = bid / tiles_n;
    const int tile_n = (int)(bid - tile_m * tiles_n);
    const long long m0 = (long long)tile_m * PM;
    const int n0 = tile_n * BN;
    const int pix_bytes = p.Cin * 4;

    // producer role: thread = (row = tid >> 2, 8-channel pair c = tid & 3): the four lanes of a row read 64 contiguous
    // bytes of the hi half and 64 of the lo half of a pixel's 128-byte block line
    const int pc = tid & 3;
    const unsigned char* src0;                                // top-left tap of the pooled pixel, channel pair c, block 0
    int arow_off;                                             // byte offset of the hi chunk inside the A buffer
    {
        const unsigned ohw = (unsigned)(p.OH * p.OW), M32 = (unsigned)p.M;
        const int row = tid >> 2;
        const unsigned pix = tile_m * PM + row;
        const unsigned pp = pix < M32 ? pix : 0u;             // rows beyond M compute pixel 0 and are dropped by the epilogue
        const unsigned n = pp / ohw;
        const unsigned rem = pp - n * ohw;
        const unsigned oh = rem / (unsigned)p.OW, ow = rem - oh * (unsigned)p.OW;
        src0 = reinterpret_cast<const unsigned char*>(p.x) + (((long long)n * p.H + 2 * oh) * p.W + (VT ? ow : 2 * ow)) * pix_bytes + pc * 16;
        arow_off = row * 128 + ((pc ^ tise_lds_swz(row)) * 16);
    }
    // weights: 8-cout DMA pieces, 2 * TNW per wave and K-step (as in the default kernel: scalar base + 32-bit lane offset)
    const unsigned char* wbase = reinterpret_cast<const unsigned char*>(p.w) + (long long)n0 * p.Kpad * 4;
    unsigned pb[2 * TNW];
    int pb_off[2 * TNW];
#pragma unroll
    for (int i = 0; i < 2 * TNW; ++i) {
        const int q = wave * 2 * TNW + i;
        const int r = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ tise_lds_swz(r);
        pb[i] = (unsigned)r * (unsigned)p.Kpad * 4u + c * 16;
        pb_off[i] = A_BYTES + q * 1024;
    }
    conv_epi::Acc16 acc_main[1][TNW], acc_corr[1][TNW];
#pragma unroll
    for (int t = 0; t < TNW; ++t) { conv_epi::acc_zero(acc_main[0][t]); conv_epi::acc_zero(acc_corr[0][t]); }
    conv_epi::float4_t sc_pre = {0.f, 0.f, 0.f, 0.f}, bs_pre = {0.f, 0.f, 0.f, 0.f};
    if (tid < BN / 4) {
        sc_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.scale + n0 + 4 * tid);
        bs_pre = *reinterpret_cast<const conv_epi::float4_t*>(p.bias + n0 + 4 * tid);
    }
    // fragments (16 rows x 32 K): row (lane & 15), chunk (lane >> 4) XOR tise_lds_swz(row); the next 16 rows 2048 bytes further
    const int f16o = (lane & 15) * 128 + (((lane >> 4) ^ tise_lds_swz(lane & 15)) << 4);
    const unsigned char* fa = lds + wm * 32 * 128;
    const int row_bytes = p.W * pix_bytes;
    const int nsteps = p.Cin / CS_BK;

#define PI_B_ISSUE(BOFF)                                                                                  \
    _Pragma("unroll") for (int i = 0; i < 2 * TNW; ++i) {                                                  \
        const unsigned char* sw_ = wbase + pb[i];                                                          \
        __builtin_amdgcn_global_load_lds(sw_, (lds_ptr_t)(lds + (BOFF) + pb_off[i]), 16, 0, 0);            \
        pb[i] += 128;                                                                                      \
    }
    PI_B_ISSUE(0)
    for (int step = 0; step < nsteps; ++step) {
        const int bcur = (step & 1) * B_BYTES;
        // ---- pooled pixel operand of this K-step: 9 taps x (hi, lo) 16-byte loads, all in flight together ----------
        half8_t ph, pl;
        {
            const unsigned char* s0 = src0 + step * 128;
            constexpr int NDW = VT ? 1 : 3, NTAP = 3 * NDW;
            half8_t vh[NTAP], vl[NTAP];
#pragma unroll
            for (int dh = 0; dh < 3; ++dh)
#pragma unroll
                for (int dw = 0; dw < NDW; ++dw) {
                    const unsigned char* q = s0 + dh * row_bytes + dw * pix_bytes;
                    vh[dh * NDW + dw] = *reinterpret_cast<const half8_t*>(q);
                    vl[dh * NDW + dw] = *reinterpret_cast<const half8_t*>(q + 64);
                }
            float bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) bv[i] = -INFINITY;
#pragma unroll
            for (int t = 0; t < NTAP; ++t)
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = fmaxf(bv[i], (float)vh[t][i] + (float)vl[t][i] * (1.f / 2048.f));
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                ph[i] = (_Float16)bv[i];
                pl[i] = (_Float16)((bv[i] - (float)ph[i]) * 2048.f);
            }
        }
        __syncthreads();                                      // the previous step's fragment reads of the A buffer are done
        *reinterpret_cast<half8_t*>(lds + arow_off) = ph;
        *reinterpret_cast<half8_t*>(lds + (arow_off ^ 64)) = pl;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this step's weights have landed (issued one step ago)
        __syncthreads();
        // next step's weights into the other stage (last read by the MFMAs of step - 1, which ended before the first
        // barrier above); issued AFTER the barrier, whose fence would otherwise wait for them
        if (step + 1 < nsteps) PI_B_ISSUE(((step + 1) & 1) * B_BYTES)
        // ---- MFMAs: the default kernel's order (bit-identical accumulation) ---------------------------------
        half8_t a_[2][2];
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
            a_[pi][0] = *reinterpret_cast<const half8_t*>(fa + pi * 2048 + f16o);
            a_[pi][1] = *reinterpret_cast<const half8_t*>(fa + pi * 2048 + (f16o ^ 64));
        }
#pragma unroll
        for (int g = 0; g < 2 * TNW; ++g) {
            const unsigned char* bb = lds + A_BYTES + bcur + (wn * 2 * TNW + g) * 2048;
            const half8_t b_hi = *reinterpret_cast<const half8_t*>(bb + f16o);
            const half8_t b_lo = *reinterpret_cast<const half8_t*>(bb + (f16o ^ 64));
#pragma unroll
            for (int pi = 0; pi < 2; ++pi) {
                float4_t& cm = acc_main[0][g >> 1].v[g & 1][pi];
                float4_t& cc = acc_corr[0][g >> 1].v[g & 1][pi];
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_lo, a_[pi][0], cc, 0, 0, 0);
                cm = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_hi, a_[pi][0], cm, 0, 0, 0);
                cc = __builtin_amdgcn_mfma_f32_16x16x32_f16(b_hi, a_[pi][1], cc, 0, 0, 0);
            }
        }
    }
#undef PI_B_ISSUE
    __syncthreads();
    conv_epi::prepare<BN>(p, lds + EPI0, n0, sc_pre, bs_pre);
    __syncthreads();
    conv_epi::store_tiles_desc<TNW, ETW, false, BN>(p, acc_main, acc_corr, lds + wave * conv_epi::Staging<ETW>::BYTES, lds + EPI0,
                                                    m0 + wm * 32, wn * TNW * 4);
}

static int launch_poolin(const ConvArgs* a, int tn, hipStream_t st, bool vt = false) {
    if (a->KH != 1 || a->KW != 1 || a->PH != 0 || a->PW != 0 || a->Cin % 32 != 0 || a->Kpad != a->Cin || a->H < 3 || a->W < (vt ? 1 : 3) ||
        a->OH != (a->H - 3) / 2 + 1 || a->OW != (vt ? a->W : (a->W - 3) / 2 + 1) || a->M != (long long)a->N * a->OH * a->OW || a->M >= 0x7fffff00LL ||
        (long long)a->W * a->Cin * 4 * 3 >= 0x7fffffffLL)
        return TISE_ERR_INVALID_ARG;
    (void)tn;                                                 // the tile width follows from Cout: one 128-cout tile, else 256-cout tiles
    const int tnw = a->Cout <= 128 ? 2 : 4;
    const int bn = 64 * tnw;
    const long long tiles = ((a->M + 63) / 64) * ((a->Cout + bn - 1) / bn);
    if (tiles > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    const dim3 grid((unsigned)tiles), block(256);
    if (vt) {
        if (tnw == 2) hipLaunchKernelGGL((conv_poolin_kernel<2, true>), grid, block, 0, st, *a);
        else hipLaunchKernelGGL((conv_poolin_kernel<4, true>), grid, block, 0, st, *a);
    } else if (tnw == 2) hipLaunchKernelGGL(conv_poolin_kernel<2>, grid, block, 0, st, *a);
    else hipLaunchKernelGGL(conv_poolin_kernel<4>, grid, block, 0, st, *a);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_conv_pipe_launch(const tise_conv_args* a, int cfg, void* stream);   // conv_pipe.hip

extern "C" int tise_conv_split_f16(const ConvArgs* args, int tn, void* stream) {
    if (!args || !args->x || !args->w || !args->scale || !args->bias || (args->nseg & 0xff) < 1 || (args->nseg & 0xff) > 4 ||
        args->Cin % 16 != 0 || args->Cin < 32 || (args->Kpad % CS_BK != 0 && !(tn & 512)) || args->M <= 0)
        return TISE_ERR_INVALID_ARG;
    // the epilogues work on 8-cout chunks and 16-byte stores: segments must start on multiples of 8 couts and land
    // on 16-byte boundaries (split tensors: 8 channels inside a 16- or 32-channel block, fp32: 4 elements)
    for (int i = 0; i < (args->nseg & 0xff); ++i) {
        const tise_conv_seg& g = args->seg[i];
        const int al = g.mode == 0 ? 8 : 4;
        if (!g.dst || g.c0 % 8 != 0 || g.c1 <= g.c0 || g.off % al != 0 || g.ld % (g.mode == 0 ? 16 : 4) != 0 ||
            (g.mode == 0 && g.off + (g.c1 - g.c0) > g.ld) ||
            (reinterpret_cast<uintptr_t>(g.dst) & 15) != 0 || (i > 0 && g.c0 != args->seg[i - 1].c1) || (g.mode != 0 && g.mode != 1))
            return TISE_ERR_INVALID_ARG;
    }
    if (args->seg[0].c0 != 0) return TISE_ERR_INVALID_ARG;
    if (tn & 512) return tise_conv_pipe_launch(args, (tn & 255) | (tn & 1024), stream);   // resident-weights sliding-window kernel (| 1024: pooled output)
    if (args->out_hp | args->out_wp | args->out_y0 | args->out_x0) return TISE_ERR_INVALID_ARG;   // offset destinations: sliding-window kernels only
    // | 2048: the pool is split -- the producer (row-window kernel) takes the horizontal half, the consumer the vertical one
    if (tn & 256) return launch_poolin(args, tn & 15, (hipStream_t)stream, (tn & 2048) != 0);   // max-pool fused into a 1x1 convolution's operand load
    if (tn & 64) return launch_rowwin_any(args, tn & 15, (hipStream_t)stream, (tn & 2048) != 0);   // row-window kernel, K order (kh, block, kw)
    const bool glds = (tn & 16) != 0;
    // fast path: K order (tap, full 32-channel block) then paired 16-channel tails (see the kernel); Kpad says which
    const int fast_kpad = (args->KH * args->KW * (args->Cin / 32) + ((args->Cin & 16) ? (args->KH * args->KW + 1) / 2 : 0)) * 32;
    const bool fast = (tn & 128) != 0 && args->Cin % 16 == 0 && args->Kpad == fast_kpad && args->M < 0x7fffff00LL &&
                      args->H < 0x3f00 && args->W < 0x3f00 && args->PH < 0x100 && args->PW < 0x100;   // packed (ih0, iw0)
    if ((tn & 128) && !fast) return TISE_ERR_UNSUPPORTED;     // the weight packing differs: no silent fall-back
    const bool cbt = (tn & 4096) != 0;                        // weights packed (channel block, tap): conv_split.py korder="block"
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
    if (fast && cbt) {                                         // K order (channel block, tap): unpadded multi-tap layers
        if (args->PH != 0 || args->PW != 0 || args->Cin % 32 != 0 || args->KH * args->KW < 2) return TISE_ERR_INVALID_ARG;
        switch (tn) {
            case 2: hipLaunchKernelGGL((conv_split_fast_kernel<2, false, true>), grid, block, 0, st, *args); break;
            case 3: hipLaunchKernelGGL((conv_split_fast_kernel<3, false, true>), grid, block, 0, st, *args); break;
            case 4: hipLaunchKernelGGL((conv_split_fast_kernel<4, false, true>), grid, block, 0, st, *args); break;
            case 5: hipLaunchKernelGGL((conv_split_fast_kernel<5, false, true>), grid, block, 0, st, *args); break;
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
    return TISE_ERR_INVALID_ARG;                              // (the register-staged variant 0 was removed in round 2)
}

TISE_DEFINE_SPLIT_FLAG_READER(tise_internal_split_flag_conv_split)
