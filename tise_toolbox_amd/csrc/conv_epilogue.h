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
//      destination: 64-byte runs per pixel and plane (128-byte for raw fp32).
// No workgroup barrier: a wave's LDS operations execute in order, and nothing else touches its staging area.
#pragma once
#include <hip/hip_fp16.h>
#include "common.h"

namespace conv_epi {

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));


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
#define CONV_EPI_SEG(COL, DST, LD, PLANE, OFF, MODE, C0)                                                  \
    void* DST = p.seg[0].dst;                                                                             \
    long long LD = p.seg[0].ld, PLANE = p.seg[0].plane;                                                   \
    int OFF = p.seg[0].off, MODE = p.seg[0].mode, C0 = p.seg[0].c0;                                       \
    _Pragma("unroll") for (int s_ = 1; s_ < 4; ++s_)                                                       \
        if (s_ < nseg && (COL) >= p.seg[s_].c0) {                                                          \
            DST = p.seg[s_].dst; LD = p.seg[s_].ld; PLANE = p.seg[s_].plane;                               \
            OFF = p.seg[s_].off; MODE = p.seg[s_].mode; C0 = p.seg[s_].c0;                                 \
        }

// acc[i][t]: tile rows m0w + 32*i .. +31, couts n0w + 32*t .. +31.  tw: this wave's staging bytes of LDS.
// TW = accumulator tiles staged together along the couts (1 or 2): the runs stored per pixel and plane are
// TW * 64 bytes (a full 128-byte line for TW = 2, which is what the HBM-write-bound 147^2 layers need).
template <int TW>
struct Staging {
    static constexpr int PITCH = TW * 128 + 16;   // staging row: TW x 32 couts x 4 B + 16 B pad
    static constexpr int BYTES = 32 * PITCH;      // per wave
};

template <int TMW, int TNW, bool GRID, int TW>
__device__ __forceinline__ void store_tiles(const tise_conv_args& p, float16_t (&acc_main)[TMW][TNW],
                                            float16_t (&acc_corr)[TMW][TNW], unsigned char* tw, long long m0w, int n0w) {
    constexpr int PITCH = Staging<TW>::PITCH;
    const int lane = threadIdx.x & 63;
    const int nseg = p.nseg & 0xff;
#pragma unroll
    for (int i = 0; i < TMW; ++i)
#pragma unroll
        for (int t0 = 0; t0 < TNW; t0 += TW) {
            constexpr int dummy = 0; (void)dummy;
            const int nt = (TNW - t0) < TW ? (TNW - t0) : TW;        // tiles in this group (compile-time after unrolling)
            unsigned char* trow = tw + (lane & 31) * PITCH;
#pragma unroll
            for (int u = 0; u < TW; ++u) {
                if (u >= nt) continue;
                const int t = t0 + u;
                const int cb0 = n0w + t * 32;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ch = cb0 + 8 * g + 4 * (lane >> 5);
                    const float4_t sc = *reinterpret_cast<const float4_t*>(p.scale + ch);
                    const float4_t bs = *reinterpret_cast<const float4_t*>(p.bias + ch);
                    CONV_EPI_SEG(ch, sd_, sl_, sp_, so_, smode, sc0_)
                    (void)sd_; (void)sl_; (void)sp_; (void)so_; (void)sc0_;
                    float4_t v;
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        v[k] = (acc_main[i][t][4 * g + k] + acc_corr[i][t][4 * g + k] * (1.0f / 2048.0f)) * sc[k];
                    unsigned char* slot = trow + u * 128 + g * 32;
                    if (smode == 0) {
                        half4_t hi, lo;
#pragma unroll
                        for (int k = 0; k < 4; ++k) {
                            const float r = fmaxf(v[k] + bs[k], 0.f);
                            hi[k] = (_Float16)r;
                            lo[k] = (_Float16)((r - (float)hi[k]) * 2048.0f);
                        }
                        *reinterpret_cast<half4_t*>(slot + (lane >> 5) * 8) = hi;
                        *reinterpret_cast<half4_t*>(slot + 16 + (lane >> 5) * 8) = lo;
                    } else {
                        *reinterpret_cast<float4_t*>(slot + (lane >> 5) * 16) = v;
                    }
                }
            }
            // read back: 8 * nt lanes per pixel row, 16 bytes each
            const int lpr = 8 * nt;                                  // lanes per row: 8 or 16
            const int rows_per_pass = 64 / lpr;
            const int row0 = lane / lpr, q = lane % lpr;
            const int col = n0w + t0 * 32 + 8 * (q >> 1);
            CONV_EPI_SEG(col, sd_, sl_, sp_, so_, smode, sc0_)
            const bool col_ok = col < p.Cout;
            if (!GRID) {
                // tile rows are output pixels: one 64-bit address per lane, advanced by a constant per pass, and a
                // wave-uniform row limit instead of per-store pixel arithmetic
                const long long pp0 = m0w + i * 32 + row0;
                const long long left = p.M - (m0w + i * 32);          // rows of this tile that exist
                const int rows_ok = left > 32 ? 32 : (left < 0 ? 0 : (int)left);
                const long long esz = smode == 0 ? 2 : 4;
                unsigned char* d = reinterpret_cast<unsigned char*>(sd_) +
                                   (((smode == 0 && (q & 1)) ? sp_ : 0) + pp0 * sl_ + so_ + (col - sc0_) +
                                    ((smode != 0 && (q & 1)) ? 4 : 0)) * esz;
                const long long step = (long long)rows_per_pass * sl_ * esz;
                const unsigned char* src = tw + row0 * PITCH + q * 16;
#pragma unroll
                for (int r4 = 0; r4 < 32 * TW / 8; ++r4) {
                    if (r4 * rows_per_pass >= 32) continue;
                    const u32x4_t val = *reinterpret_cast<const u32x4_t*>(src + r4 * rows_per_pass * PITCH);
                    if (col_ok && row0 + r4 * rows_per_pass < rows_ok) *reinterpret_cast<u32x4_t*>(d) = val;
                    d += step;
                }
            } else {
#pragma unroll
                for (int r4 = 0; r4 < 32 * TW / 8; ++r4) {
                    if (r4 * rows_per_pass >= 32) continue;
                    const int row = r4 * rows_per_pass + row0;
                    const long long pp = out_pixel<GRID>(p, m0w + i * 32 + row);
                    const u32x4_t val = *reinterpret_cast<const u32x4_t*>(tw + row * PITCH + q * 16);
                    if (col_ok && pp >= 0) {
                        if (smode == 0) {
                            _Float16* d = reinterpret_cast<_Float16*>(sd_) + ((q & 1) ? sp_ : 0) + pp * sl_ + so_ + (col - sc0_);
                            *reinterpret_cast<u32x4_t*>(d) = val;
                        } else {
                            float* d = reinterpret_cast<float*>(sd_) + pp * sl_ + so_ + (col - sc0_) + 4 * (q & 1);
                            *reinterpret_cast<u32x4_t*>(d) = val;
                        }
                    }
                }
            }
        }
}

}  // namespace conv_epi
