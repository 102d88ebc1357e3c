// Wave-private epilogue of the split-precision convolution kernels (conv_split.hip "fast"/"glds", conv_pipe.hip).
//
// The kernels issue their MFMAs with the WEIGHT fragment as the first operand, so the 32 x 32 accumulator of a
// lane holds pixel (lane & 31) and, per register quad g = j >> 2, the four CONSECUTIVE couts
// 8*g + 4*(lane >> 5) + (j & 3).  One tile at a time the wave
//   1. applies the folded-BatchNorm scale (and bias + ReLU for split segments), re-splits v ~= hi + lo * 2^-11,
//   2. stages the tile (or two neighbouring tiles) in its own 4.5 (8.5) KB of LDS: per pixel row and 8-cout chunk
//      one 32-byte slot holding
//      [hi x8 | lo x8] (split segment, mode 0) or fp32 x8 (raw segment, mode 1) -- 8- and 16-byte LDS writes,
//   3. reads the rows back 16 bytes per lane (8 lanes = one pixel's 32 couts) and stores them to the segment's
//      destination: in the interleaved split layout (common.h) the 32 couts of a tile are ONE 128-byte line
//      [hi x32 | lo x32] of the pixel (128-byte runs for raw fp32 too).
// No workgroup barrier: a wave's LDS operations execute in order, and nothing else touches its staging area.
#pragma once
#include <hip/hip_fp16.h>
#include <type_traits>
#include "common.h"

namespace conv_epi {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));

// Accumulator of a 32-pixel x 32-cout wave tile held as four 16 x 16 blocks of v_mfma_f32_16x16x32_f16 (weight fragment
// first): v[ci][pi] = couts 16 ci .. +15 x pixels 16 pi .. +15; a lane holds pixel 16 pi + (lane & 15) and the four
// CONSECUTIVE couts 16 ci + 4 (lane >> 4) + k, i.e. half ((lane >> 4) & 1) of the tile's 8-cout chunk 2 ci + (lane >> 5).
// (float16_t = the 32 x 32 block of v_mfma_f32_32x32x16_f16: pixel (lane & 31), couts 8 g + 4 (lane >> 5) + k.)
// destination pointers come out of LDS descriptors as plain integers: tell the compiler they are GLOBAL memory, or it emits
// flat_store (address-space check per lane, counted on both vmcnt and lgkmcnt) instead of global_store (+0.3 % images/s).
// The stores are non-temporal (`nt`): an activation tensor is written once and read by a LATER kernel, by which time a
// 500-image batch has pushed it out of the 32 MB of L2 anyway -- without the hint the lines it allocates evict the weights
// and tap-shifted inputs the running kernel re-reads (+0.5 % images/s).  tools/conv_ablate.py (switch 0x2000): the stores,
// not the conversion, are the larger part of what the epilogue costs a launch (17x17x768->704: 0.58 ms with, 0.48 without
// the stores, 0.44 without any epilogue).
typedef __attribute__((address_space(1))) u32x4_t* global_u32x4_ptr;
__device__ __forceinline__ global_u32x4_ptr global_ptr(unsigned char* p) {
    return (global_u32x4_ptr)(unsigned long long)p;
}

struct Acc16 {
    float4_t v[2][2];
};
__device__ __forceinline__ void acc_zero(Acc16& a) {
#pragma unroll
    for (int i = 0; i < 4; ++i) a.v[i >> 1][i & 1] = float4_t{0.f, 0.f, 0.f, 0.f};
}
__device__ __forceinline__ void acc_zero(float16_t& a) {
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = 0.f;
}


// Tile row -> output pixel index (or -1).  GRID = false: tile rows are output pixels.  GRID = true (window
// kernels): tile rows are pixels of the INPUT grid (n, y, x); the output pixel is (n, y, x) when it exists.
template <bool GRID>
__device__ __forceinline__ long long out_pixel(const tise_conv_args& p, long long g) {
    if (!GRID) return g < p.M ? g : -1;
    const long long hw = (long long)p.H * p.W;
    if (g >= (long long)p.N * hw) return -1;
    const long long n = g / hw;
    const int rem = (int)(g - n * hw);
    const int y = rem / p.W, x = rem - y * p.W;
    if (y >= p.OH || x >= p.OW) return -1;
    return (n * p.OH + y) * p.OW + x;
}

// static-index segment look-up (dynamic indexing of the by-value argument struct would spill it to scratch)
#define CONV_EPI_SEG(COL, DST, LD, OFF, MODE, C0)                                                         \
    void* DST = p.seg[0].dst;                                                                             \
    long long LD = p.seg[0].ld;                                                                           \
    int OFF = p.seg[0].off, MODE = p.seg[0].mode, C0 = p.seg[0].c0;                                       \
    _Pragma("unroll") for (int s_ = 1; s_ < 4; ++s_)                                                       \
        if (s_ < nseg && (COL) >= p.seg[s_].c0) {                                                          \
            DST = p.seg[s_].dst; LD = p.seg[s_].ld;                                                        \
            OFF = p.seg[s_].off; MODE = p.seg[s_].mode; C0 = p.seg[s_].c0;                                 \
        }

// acc[i][t]: tile rows m0w + 32*i .. +31, couts n0w + 32*t .. +31.  tw: this wave's staging bytes of LDS.
// TW = accumulator tiles staged together along the couts (1 or 2): a store pass writes TW * 128 contiguous bytes per
// pixel (one [hi x32 | lo x32] line per tile).
template <int TW>
struct Staging {
    static constexpr int PITCH = TW * 128 + 16;   // staging row: TW x 32 couts x 4 B + 16 B pad
    static constexpr int BYTES = 32 * PITCH;      // per wave
};

// ------------------------------------------------------------------------------------------------
// Descriptor form.  Looking the destination segment up with per-lane select chains and loading scale / bias from
// global memory inside the tile loop (round 1's first epilogue) showed one `s_waitcnt vmcnt(0)` per 8-cout group
// in the ISA (each also waiting for the previous group's stores) and 30 exec-mask branches per tile.  Here the
// workgroup first writes, once, into LDS:
//   * scale[BN], bias[BN] (prefetched into registers before the K loop),
//   * one 32-byte descriptor per 8-cout chunk: {byte address of (pixel 0, first cout of the chunk, hi half),
//     bytes per pixel, byte offset of the second half (lo half of the block / couts 4..7), mode, valid},
// and the tile loop only reads LDS; the segment mode of a group is wave-uniform (segments start on multiples
// of 8 couts), so it is taken through readfirstlane and the split / raw paths become scalar branches.
struct ChunkDesc {
    long long base, row_stride, second;
    int mode, valid;
};

template <int BN>
struct EpiArea {
    static constexpr int DESC = 0, SCALE = (BN / 8) * 32, BIAS = SCALE + BN * 4, BYTES = BIAS + BN * 4;
};

// all threads of the workgroup; area = LDS behind the staging tiles.  sc_pre / bs_pre: scale / bias of couts
// n0 + 4*tid .. +3 held by threads tid < BN / 4.  Caller issues __syncthreads() afterwards.
template <int BN>
__device__ __forceinline__ void prepare(const tise_conv_args& p, unsigned char* area, int n0, float4_t sc_pre, float4_t bs_pre,
                                        int tid = -1) {
    if (tid < 0) tid = threadIdx.x;
    const int nseg = p.nseg & 0xff;
    if (tid < BN / 4) {
        *reinterpret_cast<float4_t*>(area + EpiArea<BN>::SCALE + tid * 16) = sc_pre;
        *reinterpret_cast<float4_t*>(area + EpiArea<BN>::BIAS + tid * 16) = bs_pre;
    }
    if (tid < BN / 8) {
        const int col = n0 + 8 * tid;
        CONV_EPI_SEG(col, sd_, sl_, so_, smode, sc0_)
        const int ch = so_ + (col - sc0_);                     // destination channel
        ChunkDesc d;
        d.base = (long long)reinterpret_cast<unsigned char*>(sd_) + (smode == 0 ? tise_ilv_off(ch, (int)sl_) * 2 : ch * 4);
        d.row_stride = sl_ * 4;                                // split pixel = 4*C bytes, the same as fp32
        if (p.out_hp) d.base += ((long long)p.out_y0 * p.out_wp + p.out_x0) * d.row_stride;      // result at an offset inside a larger image (GRID kernels)
        d.second = smode == 0 ? tise_ilv_second(ch, (int)sl_) * 2 : 16;
        d.mode = smode;
        d.valid = col < p.Cout ? 1 : 0;
        *reinterpret_cast<ChunkDesc*>(area + EpiArea<BN>::DESC + tid * 32) = d;
    }
}

// Four output values of a split-format segment -> (hi x4, lo x4) (round 6, VERDICT r5 weak 1: the epilogue spent ~7.5 vector
// operations per value -- profiles/r06g_epilogue_isa.txt counts them in the compiler's code of round 5's expressions):
//     r  = max(v * scale + bias, 0)          ONE fma per value, packed in pairs (v = main + corr / 2048 is one fma too; round 5: add, mul, add)
//     hi = fp16(r)                           ONE v_cvt_pk_f16_f32 per pair (round 5's code converted every hi twice: once alone for
//                                            the residual, once packed for the store)
//     lo = fp16((r - hi) * 2048)             r - hi is exact in fp32 and is formed by a mixed-precision fma that reads hi AS fp16
//                                            straight from the packed register (v_fma_mix_f32; round 5: convert back, subtract);
//                                            the scaling by 2048 is exact; one packed conversion per pair
// = ~5.75 operations per value with the ReLU, the range guard's running maximum and the corr / 2048 fma.  vmax keeps the running maximum for the range guard (common.h).  The fused multiply-add rounds once
// where round 5 rounded twice: results differ from round 5's in the last bit of the 22-bit value, never by more.
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split_store_pair(float v0, float v1, float sc0, float sc1, float bs0, float bs1, float& vmax,
                                                 half2_t& hi, half2_t& lo) {
    const float r0 = fmaxf(__builtin_fmaf(v0, sc0, bs0), 0.f), r1 = fmaxf(__builtin_fmaf(v1, sc1, bs1), 0.f);
    vmax = fmaxf(vmax, fmaxf(r0, r1));
    const float2_t r = {r0, r1};
    hi = __builtin_convertvector(r, half2_t);
    const unsigned hpk = __builtin_bit_cast(unsigned, hi);
    float t0, t1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(hpk), "v"(r0));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(hpk), "v"(r1));
    const float2_t t = {t0 * 2048.0f, t1 * 2048.0f};
    lo = __builtin_convertvector(t, half2_t);
}
__device__ __forceinline__ void split_store_quad(const float4_t& v, const float4_t& sc, const float4_t& bs, float& vmax, half4_t& hi, half4_t& lo) {
    half2_t h0, l0, h1, l1;
    split_store_pair(v[0], v[1], sc[0], sc[1], bs[0], bs[1], vmax, h0, l0);
    split_store_pair(v[2], v[3], sc[2], sc[3], bs[2], bs[3], vmax, h1, l1);
    hi = half4_t{h0[0], h0[1], h1[0], h1[1]};
    lo = half4_t{l0[0], l0[1], l1[0], l1[1]};
}

// acc[0][t]: tile rows m0w .. m0w+31 (this wave), couts n0 + 32*t .. +31.  tw: the wave's staging bytes.
// GRID = true (window kernels): tile rows are pixels of the input grid; the (n, y, x) of a lane's first row comes
// from two 32-bit divisions and is stepped forward per pass (grid pixel counts < 2^31: launcher).
// BN_AREA: tile width the descriptor area was prepared for (default: this wave's TNW tiles are the whole tile);
// chunk0: first 8-cout chunk of this wave inside the tile (wave layouts with more than one wave along the couts).
// nyx (GRID only): {n, y, x} of grid pixel m0w + (lane / lanes-per-row), when the caller tracks it incrementally.
template <int TNW, int TW, bool GRID = false, int BN_AREA = 32 * TNW, class ACC = float16_t>
__device__ __forceinline__ void store_tiles_desc(const tise_conv_args& p, ACC (&acc_main)[1][TNW],
                                                 ACC (&acc_corr)[1][TNW], unsigned char* tw,
                                                 const unsigned char* area, long long m0w, int chunk0 = 0,
                                                 const unsigned* nyx = nullptr) {
    constexpr bool L16 = std::is_same<ACC, Acc16>::value;        // accumulator layout (see Acc16)
    constexpr int BN = BN_AREA;
    constexpr int PITCH = Staging<TW>::PITCH;
    const int lane = threadIdx.x & 63;
    const long long left = (GRID ? (long long)p.N * p.H * p.W : p.M) - m0w;   // rows of this tile that exist (wave-uniform)
    const int rows_ok = left > 32 ? 32 : (left < 0 ? 0 : (int)left);
    float vmax = 0.f;                                            // range guard of the split format (common.h)
#pragma unroll
    for (int t0 = 0; t0 < TNW; t0 += TW) {
        const int nt = (TNW - t0) < TW ? (TNW - t0) : TW;
        if constexpr (!L16) {
            unsigned char* trow = tw + (lane & 31) * PITCH;
            // mode, scale and bias of the group's 4 * nt chunks are requested in ONE batch before any of them is used: read
            // chunk by chunk inside the conversion loop, every chunk exposed two or three LDS round trips (scale, mode,
            // bias behind the mode branch) -- some 300 cycles x 16 chunks per wave and tile at TN = 4, the larger part of the
            // epilogue's 10-30 % share of a launch (profiles/r01g_conv_ablation.txt)
            int modes[TW * 4];
            float4_t scs[TW * 4], bss[TW * 4];
    #pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u >= nt) continue;
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int chunk = chunk0 + (t0 + u) * 4 + g;
                    const int ch = chunk * 8 + 4 * (lane >> 5);
                    modes[u * 4 + g] = *reinterpret_cast<const int*>(area + EpiArea<BN>::DESC + chunk * 32 + 24);
                    scs[u * 4 + g] = *reinterpret_cast<const float4_t*>(area + EpiArea<BN>::SCALE + ch * 4);
                    bss[u * 4 + g] = *reinterpret_cast<const float4_t*>(area + EpiArea<BN>::BIAS + ch * 4);
                }
            }
    #pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u >= nt) continue;
                const int t = t0 + u;
    #pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int mode = __builtin_amdgcn_readfirstlane(modes[u * 4 + g]);
                    const float4_t sc = scs[u * 4 + g];
                    float4_t v;
    #pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[k] = __builtin_fmaf(acc_corr[0][t][4 * g + k], 1.0f / 2048.0f, acc_main[0][t][4 * g + k]);
                    unsigned char* slot = trow + u * 128 + g * 32;
                    if (mode == 0) {
                        const float4_t bs = bss[u * 4 + g];
                        half4_t hi, lo;
                        split_store_quad(v, sc, bs, vmax, hi, lo);
                        *reinterpret_cast<half4_t*>(slot + (lane >> 5) * 8) = hi;
                        *reinterpret_cast<half4_t*>(slot + 16 + (lane >> 5) * 8) = lo;
                    } else {
    #pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] *= sc[k];
                        *reinterpret_cast<float4_t*>(slot + (lane >> 5) * 16) = v;
                    }
                }
            }
        } else {
            // 16 x 16 blocks: per tile and cout half ci a lane owns ONE half-chunk (4 couts) of chunk 2 ci + (lane >> 5), for the
            // two pixels (lane & 15) and 16 + (lane & 15).  Segments start on multiples of 8 couts, so the two chunks a wave
            // converts together may belong to segments of different modes: the mode is per lane (exec-masked branches)
            unsigned char* trow = tw + (lane & 15) * PITCH;
            const int sub = (lane >> 4) & 1;
            int modes[TW * 2];
            float4_t scs[TW * 2], bss[TW * 2];
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u >= nt) continue;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int chunk = chunk0 + (t0 + u) * 4 + ci * 2 + (lane >> 5);
                    const int ch = chunk * 8 + 4 * sub;
                    modes[u * 2 + ci] = *reinterpret_cast<const int*>(area + EpiArea<BN>::DESC + chunk * 32 + 24);
                    scs[u * 2 + ci] = *reinterpret_cast<const float4_t*>(area + EpiArea<BN>::SCALE + ch * 4);
                    bss[u * 2 + ci] = *reinterpret_cast<const float4_t*>(area + EpiArea<BN>::BIAS + ch * 4);
                }
            }
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u >= nt) continue;
                const int t = t0 + u;
#pragma unroll
                for (int ci = 0; ci < 2; ++ci) {
                    const int mode = modes[u * 2 + ci];
                    const float4_t sc = scs[u * 2 + ci], bs = bss[u * 2 + ci];
#pragma unroll
                    for (int pi = 0; pi < 2; ++pi) {
                        float4_t v;
#pragma unroll
                        for (int k = 0; k < 4; ++k) v[k] = __builtin_fmaf(acc_corr[0][t].v[ci][pi][k], 1.0f / 2048.0f, acc_main[0][t].v[ci][pi][k]);
                        unsigned char* slot = trow + pi * 16 * PITCH + u * 128 + (ci * 2 + (lane >> 5)) * 32;
                        if (mode == 0) {
                            half4_t hi, lo;
                            split_store_quad(v, sc, bs, vmax, hi, lo);
                            *reinterpret_cast<half4_t*>(slot + sub * 8) = hi;
                            *reinterpret_cast<half4_t*>(slot + 16 + sub * 8) = lo;
                        } else {
#pragma unroll
                            for (int k = 0; k < 4; ++k) v[k] *= sc[k];
                            *reinterpret_cast<float4_t*>(slot + sub * 16) = v;
                        }
                    }
                }
            }
        }
        const int lpr = 8 * nt;                               // lanes per pixel row: 8 or 16
        const int rows_per_pass = 64 / lpr;
        const int row0 = lane / lpr, q = lane % lpr;
        const ChunkDesc cd = *reinterpret_cast<const ChunkDesc*>(area + EpiArea<BN>::DESC + (chunk0 + t0 * 4 + (q >> 1)) * 32);
        const unsigned char* src = tw + row0 * PITCH + q * 16;
        // all staged rows of the group are read back in one batch, THEN stored (a read right in front of each
        // conditional store exposed one LDS round trip per pass)
        constexpr int NPASS = 32 * TW / 8;
        u32x4_t vals[NPASS];
#pragma unroll
        for (int r4 = 0; r4 < NPASS; ++r4) {
            if (r4 * rows_per_pass >= 32) continue;
            vals[r4] = *reinterpret_cast<const u32x4_t*>(src + r4 * rows_per_pass * PITCH);
        }
        if (!GRID) {
            unsigned char* d = reinterpret_cast<unsigned char*>(cd.base + (m0w + row0) * cd.row_stride + ((q & 1) ? cd.second : 0));
            const long long step = rows_per_pass * cd.row_stride;
#pragma unroll
            for (int r4 = 0; r4 < NPASS; ++r4) {
                if (r4 * rows_per_pass >= 32) continue;
                if (cd.valid && row0 + r4 * rows_per_pass < rows_ok && !(p.nseg & 0x2000)) __builtin_nontemporal_store(vals[r4], global_ptr(d));
                d += step;
            }
        } else {
            // (n, y, x) of grid pixel m0w + row0: two 32-bit divisions, or handed in by a kernel that tracks them
            unsigned n, y, x;
            if (nyx) { n = nyx[0]; y = nyx[1]; x = nyx[2]; }
            else {
                const unsigned g0 = (unsigned)(m0w + row0);
                const unsigned hw = (unsigned)(p.H * p.W);
                n = g0 / hw;
                const unsigned rem = g0 - n * hw;
                y = rem / (unsigned)p.W; x = rem - y * (unsigned)p.W;
            }
            const long long half_off = (q & 1) ? cd.second : 0;
            const int ohp = p.out_hp ? p.out_hp : p.OH, owp = p.out_hp ? p.out_wp : p.OW;      // destination image pitch
            // the destination address of (n, y, x) is built ONCE and then follows the coordinates: + rows_per_pass pixels per
            // pass, + the pitch difference when x wraps into the next image row, + the rows between two images when y wraps.
            // (Rebuilding ((n * ohp + y) * owp + x) * stride per pass cost eight quarter-rate integer multiplies per pass:
            // a third of the register-weights kernel's epilogue.)  The increments fit 32 bits and the destination pixel index
            // 32 bits unsigned (tise_conv_pipe_launch checks N * out_hp * out_wp and the strides).
            const int stride = (int)cd.row_stride;
            unsigned char* d = reinterpret_cast<unsigned char*>(cd.base) +
                               ((unsigned long long)((n * (unsigned)ohp + y) * (unsigned)owp + x)) * (unsigned)stride + half_off;
            const int sx = rows_per_pass * stride;                // signed: the destination image may be SMALLER than the grid
            const int sy = (owp - p.W) * stride;
            const int sn = (ohp - p.H) * owp * stride;
#pragma unroll
            for (int r4 = 0; r4 < NPASS; ++r4) {
                if (r4 * rows_per_pass >= 32) continue;
                const bool ok = cd.valid && row0 + r4 * rows_per_pass < rows_ok && y < (unsigned)p.OH && x < (unsigned)p.OW && !(p.nseg & 0x2000);
                if (ok) __builtin_nontemporal_store(vals[r4], global_ptr(d));
                x += rows_per_pass;                               // next pass: rows_per_pass grid pixels further (W >= 8)
                int adv = sx;
                if (x >= (unsigned)p.W) {
                    x -= (unsigned)p.W; adv += sy;
                    if (++y == (unsigned)p.H) { y = 0; adv += sn; }
                }
                d += (long long)adv;
            }
        }
    }
    tise_flag_split_overflow(vmax);
}

}  // namespace conv_epi
