// Fused epilogues for the InceptionV3 trunk (channels-last fp32), HBM-bound.
//
// The convolutions themselves run in MIOpen (north_star assigns the conv stack to PyTorch-ROCm).
// rocprofv3 of the first version (profiles/r01a_*) showed that 35 % of a step was NOT convolution:
// per conv one bias-add pass and one ReLU pass over the output (PyTorch adds the bias of a MIOpen
// conv as a separate elementwise kernel), torch.cat copies of every branch, and a slow NHWC
// avg_pool2d over the full-width block input.  These kernels do that work in ONE pass per tensor:
//
//   bias_relu_nhwc        out[p][off+c] = max(x[p][xoff+c] + b[c], 0)   -- reads a channel slice of a
//                         (possibly wider, fused-1x1) raw conv output and writes straight into the
//                         channel slice of the block's concatenated output (no torch.cat), or in place.
//   avgpool3_bias_relu    3x3 / stride 1 / pad 1 / count_include_pad average, + bias, ReLU.  Used AFTER
//                         the 1x1 conv of the pool branch: avg-pool and a 1x1 convolution are both
//                         linear and commute (also at the zero-padded border), so the pool runs on
//                         32..192 channels instead of 192..2048 (reference graph order: pool, then conv,
//                         torchvision InceptionA/C/E branch_pool).
//   maxpool3s2            3x3 / stride 2 max pool, optionally max(. + b, 0) first (ReLU and max commute),
//                         writing into a concat slice (InceptionB/D pool branch) or a packed tensor (stem).
//
// All are float4-vectorised along the channel dimension (every channel count / offset in the
// network is a multiple of 16).  Algorithmic bytes: one read + one write of the tensor (+8 neighbour
// reads served by L2 for the pools).
#include <stdlib.h>
#include "common.h"

namespace {

__device__ __forceinline__ float4 f4_bias_relu(float4 v, float4 b) {
    v.x = fmaxf(v.x + b.x, 0.f); v.y = fmaxf(v.y + b.y, 0.f);
    v.z = fmaxf(v.z + b.z, 0.f); v.w = fmaxf(v.w + b.w, 0.f);
    return v;
}

// grid.x over (pixel, c4) pairs; C4 = C/4
__global__ __launch_bounds__(256) void bias_relu_nhwc_kernel(const float* __restrict__ x, int64_t x_ld, int x_off,
                                                             int64_t pixels, int C4, const float* __restrict__ bias,
                                                             float* __restrict__ out, int64_t out_ld, int out_off) {
    const int64_t total = pixels * C4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / C4;
        const int c4 = (int)(e - p * C4);
        const float4 v = *reinterpret_cast<const float4*>(x + p * x_ld + x_off + 4 * c4);
        const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c4);
        *reinterpret_cast<float4*>(out + p * out_ld + out_off + 4 * c4) = f4_bias_relu(v, b);
    }
}

__global__ __launch_bounds__(256) void avgpool3_bias_relu_nhwc_kernel(const float* __restrict__ x, int64_t x_ld,
                                                                      int x_off, int N, int H, int W, int C4,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ out, int64_t out_ld,
                                                                      int out_off) {
    const int64_t total = (int64_t)N * H * W * C4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / C4;
        const int c4 = (int)(e - p * C4);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int hh = h + dh;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int ww = w + dw;
                if (ww < 0 || ww >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + (p + (int64_t)dh * W + dw) * x_ld + x_off + 4 * c4);
                s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
            }
        }
        s.x /= 9.f; s.y /= 9.f; s.z /= 9.f; s.w /= 9.f;      // count_include_pad=True: always / 9
        const float4 b = *reinterpret_cast<const float4*>(bias + 4 * c4);
        *reinterpret_cast<float4*>(out + p * out_ld + out_off + 4 * c4) = f4_bias_relu(s, b);
    }
}

template <bool BIAS_RELU>
__global__ __launch_bounds__(256) void maxpool3s2_nhwc_kernel(const float* __restrict__ x, int64_t x_ld, int x_off,
                                                              int N, int H, int W, int C4,
                                                              const float* __restrict__ bias, float* __restrict__ out,
                                                              int64_t out_ld, int out_off) {
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * C4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / C4;
        const int c4 = (int)(e - p * C4);
        const int ow = (int)(p % OW);
        const int oh = (int)((p / OW) % OH);
        const int64_t n = p / ((int64_t)OW * OH);
        const int64_t base = (n * H + 2 * oh) * W + 2 * ow;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const float4 v = *reinterpret_cast<const float4*>(x + (base + (int64_t)dh * W + dw) * x_ld + x_off + 4 * c4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        if (BIAS_RELU) m = f4_bias_relu(m, *reinterpret_cast<const float4*>(bias + 4 * c4));
        *reinterpret_cast<float4*>(out + p * out_ld + out_off + 4 * c4) = m;
    }
}

// ------------------------------------------------------------------------------------------------
// Split-fp16 variants (activations v ~= hi + lo * 2^-11 in the interleaved layout of common.h: per pixel and
// 32-channel block one 128-byte line [hi x32 | lo x32]; conv_split.hip).  All channel counts / offsets these kernels
// see are multiples of 8 that do not straddle a block half, so the hi and the lo run of a thread's channels are
// contiguous and tise_ilv_off / tise_ilv_second give their places.
typedef _Float16 half8v __attribute__((ext_vector_type(8)));
typedef float float2v __attribute__((ext_vector_type(2)));

// fp32 raw 1x1-conv output (slice) -> 3x3/s1/p1 average (count_include_pad) + bias, ReLU -> split tensor slice.
// Thread = (pixel, 8 channels): two 16-byte loads per tap, one 16-byte store per half (the pool branch's channel
// counts are 32, 64 and 192).  Summation order per channel: (dh, dw) row-major, as the 4-channel form had.
__global__ __launch_bounds__(256) void avgpool3_bias_relu_split_kernel(const float* __restrict__ x, int64_t x_ld,
                                                                       int x_off, int N, int H, int W, int C8,
                                                                       const float* __restrict__ bias,
                                                                       _Float16* __restrict__ out, int out_C,
                                                                       int out_off) {
    const int64_t total = (int64_t)N * H * W * C8;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e / C8;
        const int c8 = (int)(e - p * C8);
        const int w = (int)(p % W);
        const int h = (int)((p / W) % H);
        float4 s0 = make_float4(0.f, 0.f, 0.f, 0.f), s1 = s0;
#pragma unroll
        for (int dh = -1; dh <= 1; ++dh) {
            const int hh = h + dh;
            if (hh < 0 || hh >= H) continue;
#pragma unroll
            for (int dw = -1; dw <= 1; ++dw) {
                const int ww = w + dw;
                if (ww < 0 || ww >= W) continue;
                const float4* q = reinterpret_cast<const float4*>(x + (p + (int64_t)dh * W + dw) * x_ld + x_off + 8 * c8);
                const float4 v0 = q[0], v1 = q[1];
                s0.x += v0.x; s0.y += v0.y; s0.z += v0.z; s0.w += v0.w;
                s1.x += v1.x; s1.y += v1.y; s1.z += v1.z; s1.w += v1.w;
            }
        }
        s0.x /= 9.f; s0.y /= 9.f; s0.z /= 9.f; s0.w /= 9.f;      // count_include_pad=True: always / 9
        s1.x /= 9.f; s1.y /= 9.f; s1.z /= 9.f; s1.w /= 9.f;
        const float4 r0 = f4_bias_relu(s0, *reinterpret_cast<const float4*>(bias + 8 * c8));
        const float4 r1 = f4_bias_relu(s1, *reinterpret_cast<const float4*>(bias + 8 * c8 + 4));
        const float v[8] = {r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, r1.w};
        half8v hi, lo;
        float vmax = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            vmax = fmaxf(vmax, v[i]);
            hi[i] = (_Float16)v[i];
            lo[i] = (_Float16)((v[i] - (float)hi[i]) * 2048.f);
        }
        tise_flag_split_overflow(vmax);
        const int ch = out_off + 8 * c8;
        _Float16* d = out + p * (2 * (int64_t)out_C) + tise_ilv_off(ch, out_C);
        *reinterpret_cast<half8v*>(d) = hi;
        *reinterpret_cast<half8v*>(d + tise_ilv_second(ch, out_C)) = lo;
    }
}

// Column-walking form of the kernel above (round 3; the default): thread = (image, column x, 8 channels) walks DOWN the
// H rows keeping the horizontal 3-tap sums of the last three rows in registers, so an output costs 3 new taps (6 x 16-byte
// loads) instead of 9 (18 loads), and the index arithmetic is 32-bit and done once per thread, not per output
// (the per-output form reached 2.6-3.3 TB/s of algorithmic bytes: bound by its 18 loads and 64-bit divisions, not by
// HBM).  Summation order per channel: ((left + centre) + right) per row, then (above + this) + below.
__global__ __launch_bounds__(256) void avgpool3_bias_relu_split_colwalk_kernel(const float* __restrict__ x, int x_ld, int x_off,
                                                                               int N, int H, int W, int C8,
                                                                               const float* __restrict__ bias,
                                                                               _Float16* __restrict__ out, int out_C, int out_off) {
    const unsigned total = (unsigned)N * (unsigned)W * (unsigned)C8;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= total) return;
    const unsigned c8 = e % (unsigned)C8;
    const unsigned col = e / (unsigned)C8;
    const unsigned w = col % (unsigned)W, n = col / (unsigned)W;
    const bool hasl = w > 0, hasr = w + 1 < (unsigned)W;
    const float* base = x + ((size_t)n * H * W + w) * (size_t)x_ld + x_off + 8 * c8;
    const size_t rstride = (size_t)W * x_ld;
    const float4 b0 = *reinterpret_cast<const float4*>(bias + 8 * c8), b1 = *reinterpret_cast<const float4*>(bias + 8 * c8 + 4);
    const int ch = out_off + 8 * (int)c8;
    _Float16* obase = out + ((size_t)n * H * W + w) * (size_t)(2 * out_C) + tise_ilv_off(ch, out_C);
    const int osecond = tise_ilv_second(ch, out_C);
    const size_t ostride = (size_t)W * 2 * out_C;
    // raw taps (left, centre, right; 8 channels each) of one input row, and their horizontal sum
    struct Taps { float4 l0, l1, c0, c1, r0, r1; };
#define AP_LOAD(T, ROW)                                                                                    \
    {                                                                                                      \
        const float4* q = reinterpret_cast<const float4*>(base + (size_t)(ROW) * rstride);                  \
        T.c0 = q[0]; T.c1 = q[1];                                                                           \
        T.l0 = T.l1 = T.r0 = T.r1 = make_float4(0.f, 0.f, 0.f, 0.f);                                        \
        if (hasl) { const float4* ql = reinterpret_cast<const float4*>(base + (size_t)(ROW) * rstride - x_ld); T.l0 = ql[0]; T.l1 = ql[1]; } \
        if (hasr) { const float4* qr = reinterpret_cast<const float4*>(base + (size_t)(ROW) * rstride + x_ld); T.r0 = qr[0]; T.r1 = qr[1]; } \
    }
#define AP_HSUM(DST, T)                                                                                    \
    {                                                                                                      \
        DST[0] = (T.l0.x + T.c0.x) + T.r0.x; DST[1] = (T.l0.y + T.c0.y) + T.r0.y; DST[2] = (T.l0.z + T.c0.z) + T.r0.z; DST[3] = (T.l0.w + T.c0.w) + T.r0.w; \
        DST[4] = (T.l1.x + T.c1.x) + T.r1.x; DST[5] = (T.l1.y + T.c1.y) + T.r1.y; DST[6] = (T.l1.z + T.c1.z) + T.r1.z; DST[7] = (T.l1.w + T.c1.w) + T.r1.w; \
    }
    float hs[3][8];                                           // horizontal sums of rows y - 1, y, y + 1 (rotating)
    Taps tn, tnn;                                             // raw taps of rows y + 1 and y + 2: requested one row ahead of their use
    tn.l0 = tn.l1 = tn.c0 = tn.c1 = tn.r0 = tn.r1 = make_float4(0.f, 0.f, 0.f, 0.f);
    tnn = tn;
#pragma unroll
    for (int i = 0; i < 8; ++i) hs[0][i] = 0.f;               // the row above the image
    {
        Taps t0;
        AP_LOAD(t0, 0)
        if (H > 1) AP_LOAD(tn, 1)
        AP_HSUM(hs[1], t0)
    }
    float vmax = 0.f;
    const float bs[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    for (int y = 0; y < H; ++y) {
        if (y + 2 < H) AP_LOAD(tnn, y + 2)                    // in flight while row y is finished below
        if (y + 1 < H) { AP_HSUM(hs[2], tn) }
        else {
#pragma unroll
            for (int i = 0; i < 8; ++i) hs[2][i] = 0.f;
        }
        half8v hi, lo;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float v = fmaxf(((hs[0][i] + hs[1][i]) + hs[2][i]) / 9.f + bs[i], 0.f);      // count_include_pad=True: always / 9
            vmax = fmaxf(vmax, v);
            hi[i] = (_Float16)v;
            lo[i] = (_Float16)((v - (float)hi[i]) * 2048.f);
        }
        _Float16* d = obase + (size_t)y * ostride;
        *reinterpret_cast<half8v*>(d) = hi;                      // (non-temporal stores: 20 % faster launched alone, 0.1 ms per 1000 images SLOWER
        *reinterpret_cast<half8v*>(d + osecond) = lo;            //  inside the trunk, where the next kernel reads the concat buffer straight back)
#pragma unroll
        for (int i = 0; i < 8; ++i) { hs[0][i] = hs[1][i]; hs[1][i] = hs[2][i]; }
        tn = tnn;
    }
#undef AP_LOAD
#undef AP_HSUM
    tise_flag_split_overflow(vmax);
}

// split tensor -> 3x3 / stride 2 max pool -> split tensor slice (8 channels = 2 x 16 B per thread).  The value
// v = hi + lo * 2^-11 is exact in fp32 (two 11-bit mantissas), so the maximum is taken on it -- one mixed-precision FMA
// and one v_max per tap and channel -- and split again at the end: fp16(v) gives hi back (v lies within half an ulp of
// it; on an exact tie the neighbouring fp16 value with the complementary lo, the same v).  The first version carried the
// winning (hi, lo) pair through the taps with compare + select chains: ~500 vector instructions per thread against
// ~190 now: 5.0-5.3 instead of 4.7-4.9 TB/s launched back to back (tools/pool_probe.py); inside the trunk the launches
// take the same time as before (2.9 ms per 1000 images for the four of them), i.e. the memory system sets it there.
__global__ __launch_bounds__(256) void maxpool3s2_split_kernel(const _Float16* __restrict__ x, int x_C, int x_off,
                                                               int N, int H, int W, int C8,
                                                               _Float16* __restrict__ out, int out_C, int out_off) {
    // one (pooled pixel, 8 channels) element per thread, image = blockIdx.y: two 32-bit divisions per thread (the
    // grid-stride form spent ~330 of its ~560 vector instructions per element on four 64-bit divisions)
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const unsigned per_img = (unsigned)OH * (unsigned)OW * (unsigned)C8;
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= per_img) return;
    const unsigned pix = e / (unsigned)C8;
    const int c8 = (int)(e - pix * (unsigned)C8);
    const unsigned oh = pix / (unsigned)OW, ow = pix - oh * (unsigned)OW;
    const int64_t n = blockIdx.y;
    const int64_t p = n * OH * OW + pix;
    {
        const int64_t base = (n * H + 2 * oh) * W + 2 * ow;
        const int xo = tise_ilv_off(x_off + 8 * c8, x_C), xs = tise_ilv_second(x_off + 8 * c8, x_C);
        float bv[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) bv[i] = -INFINITY;
#pragma unroll
        for (int dh = 0; dh < 3; ++dh)
#pragma unroll
            for (int dw = 0; dw < 3; ++dw) {
                const _Float16* q = x + (base + (int64_t)dh * W + dw) * (2 * (int64_t)x_C) + xo;
                const half8v vh = *reinterpret_cast<const half8v*>(q);
                const half8v vl = *reinterpret_cast<const half8v*>(q + xs);
#pragma unroll
                for (int i = 0; i < 8; ++i) bv[i] = fmaxf(bv[i], (float)vh[i] + (float)vl[i] * (1.f / 2048.f));
            }
        half8v bh, bl;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            bh[i] = (_Float16)bv[i];
            bl[i] = (_Float16)((bv[i] - (float)bh[i]) * 2048.f);
        }
        const int ch = out_off + 8 * c8;
        _Float16* d = out + p * (2 * (int64_t)out_C) + tise_ilv_off(ch, out_C);
        *reinterpret_cast<half8v*>(d) = bh;                      // (non-temporal stores measured 1 % slower here)
        *reinterpret_cast<half8v*>(d + tise_ilv_second(ch, out_C)) = bl;
    }
}

// Stem convolution Conv2d_1a_3x3 (3 -> 32 channels, 3x3, stride 2, no padding; torchvision Inception3 /
// reference image_realism/FID/inception.py:60) straight from the fp32 NHWC network input, with the folded
// BatchNorm bias, ReLU and the fp16 split of the result fused.  K = 27 is far too small for the MFMA
// implicit-GEMM kernel (which needs Cin % 16 == 0), and through MIOpen this layer cost a conv launch, a
// zero-fill launch and a separate bias/ReLU/split pass; here it is one HBM-bound pass (fp32 FMA chains in
// (kh, kw, cin) order).  Thread = (output pixel, 8 output channels); four lanes share a pixel so a pixel's
// 32 channels leave as one 128-byte line [hi x32 | lo x32].  Weights: w[kh][kw][cin][cout] fp32 in LDS.
__global__ __launch_bounds__(256) void stem_conv3x3s2_split_kernel(const float* __restrict__ x, int N, int H, int W,
                                                                   const float* __restrict__ wt,   // [27][32]
                                                                   const float* __restrict__ bias, // [32]
                                                                   _Float16* __restrict__ out) {
    __shared__ float ws[27 * 32];
    for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i] = wt[i];
    __syncthreads();
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * 4;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int64_t p = e >> 2;
        const int cg = (int)(e & 3) * 8;
        const int ow = (int)(p % OW);
        const int oh = (int)((p / OW) % OH);
        const int64_t n = p / ((int64_t)OW * OH);
        const float* xp = x + ((n * H + 2 * oh) * W + 2 * ow) * 3;
        float acc[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) acc[c] = 0.f;
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const float* xr = xp + (int64_t)kh * W * 3;
#pragma unroll
            for (int t = 0; t < 9; ++t) {                    // (kw, cin) = 9 contiguous floats of the input row
                const float v = xr[t];
                const float* wr = ws + (kh * 9 + t) * 32 + cg;
#pragma unroll
                for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wr[c], acc[c]);
            }
        }
        half8v h, l;
        float vmax = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = fmaxf(acc[c] + bias[cg + c], 0.f);
            vmax = fmaxf(vmax, v);
            h[c] = (_Float16)v;
            l[c] = (_Float16)((v - (float)h[c]) * 2048.f);
        }
        tise_flag_split_overflow(vmax);
        _Float16* d = out + p * 64 + cg;
        *reinterpret_cast<half8v*>(d) = h;
        *reinterpret_cast<half8v*>(d + 32) = l;
    }
}

// The same convolution reading the Pillow-exact uint8 pixels (resize kernel, float output switched off) and applying
// the 3 x 256 input table itself: 134 MB instead of 536 MB of network input written and read per 500 images.
// The four lanes of a pixel (a DPP quad) fetch the 3 x 9 input bytes of its window ONCE between them -- lane r < 3
// reads the 9 bytes of window row r as three aligned dwords -- and pass them round with quad-permute moves: 3 vector
// memory instructions per lane instead of 27 byte loads (the kernel is bound by its 2.8 GB of output per 1000 images,
// and 29 memory instructions per 32 output bytes kept the address path busier than the data path).
__global__ __launch_bounds__(256) void stem_conv3x3s2_split_u8_kernel(const uint8_t* __restrict__ x, const float* __restrict__ lut, int N, int H, int W,
                                                                   const float* __restrict__ wt,   // [27][32]
                                                                   const float* __restrict__ bias, // [32]
                                                                   _Float16* __restrict__ out) {
    __shared__ float ws[27 * 32];
    __shared__ float lut_s[3 * 256];                         // byte -> network input value per channel (device.make_lut)
    for (int i = threadIdx.x; i < 27 * 32; i += 256) ws[i] = wt[i];
    for (int i = threadIdx.x; i < 3 * 256; i += 256) lut_s[i] = lut[i];
    __syncthreads();
    const int OH = (H - 3) / 2 + 1, OW = (W - 3) / 2 + 1;
    const int64_t total = (int64_t)N * OH * OW * 4;
    const int64_t total_r = (total + 255) & ~(int64_t)255;   // whole workgroups take part in the quad exchange
    const uint8_t* x_end = x + (int64_t)N * H * W * 3;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total_r; e += (int64_t)gridDim.x * blockDim.x) {
        const bool live = e < total;
        const int64_t p = live ? (e >> 2) : 0;
        const int sub = (int)(e & 3);
        const int cg = sub * 8;
        // pixel index < 2^31 (launcher): 32-bit divisions (the three 64-bit ones were ~200 of ~515 vector instructions)
        const unsigned p32 = (unsigned)p;
        const unsigned row32 = p32 / (unsigned)OW;
        const int ow = (int)(p32 - row32 * (unsigned)OW);
        const unsigned n32 = row32 / (unsigned)OH;
        const int oh = (int)(row32 - n32 * (unsigned)OH);
        const int64_t n = n32;
        const uint8_t* xp = x + ((n * H + 2 * oh) * W + 2 * ow) * 3;
        // this lane's window row (lane 3 repeats row 2; its copy is not used)
        const uint8_t* xr = xp + (int64_t)(sub < 3 ? sub : 2) * W * 3;
        unsigned r0, r1, r2;                                  // the row's 9 bytes: r0 = bytes 0-3, r1 = 4-7, r2 = byte 8
        {
            const uintptr_t a = reinterpret_cast<uintptr_t>(xr);
            const unsigned* q = reinterpret_cast<const unsigned*>(a & ~(uintptr_t)3);
            if (reinterpret_cast<const uint8_t*>(q) + 12 <= x_end) {
                const unsigned d0 = q[0], d1 = q[1], d2 = q[2];
                const int sh = (int)(a & 3) * 8;
                const unsigned long long lo64 = ((unsigned long long)d1 << 32) | d0, hi64 = ((unsigned long long)d2 << 32) | d1;
                r0 = (unsigned)(lo64 >> sh);
                r1 = (unsigned)(hi64 >> sh);
                r2 = (d2 >> sh) & 0xffu;
            } else {                                          // the last bytes of the tensor: no read past its end
                r0 = xr[0] | (xr[1] << 8) | (xr[2] << 16) | ((unsigned)xr[3] << 24);
                r1 = xr[4] | (xr[5] << 8) | (xr[6] << 16) | ((unsigned)xr[7] << 24);
                r2 = xr[8];
            }
        }
        // accumulators as four float pairs: v_pk_fma_f32 does two of a tap's eight FMAs per instruction (108 instead of
        // 216 per thread; each lane's result is the same IEEE FMA as before).  Launch time unchanged (1.0 ms per 1000
        // images, 2.9 TB/s of output against 6.5 TB/s for a plain fill, tools/write_bw_probe.py): what is left per thread
        // is the byte extraction, the table look-ups through LDS and the 54 weight reads, not the FMAs
        float2v acc2[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc2[c] = float2v{0.f, 0.f};
#define STEM_ROW(KH, QP)                                                                                   \
        {                                                                                                  \
            const unsigned b0 = (unsigned)__builtin_amdgcn_mov_dpp((int)r0, QP, 0xf, 0xf, true);           \
            const unsigned b1 = (unsigned)__builtin_amdgcn_mov_dpp((int)r1, QP, 0xf, 0xf, true);           \
            const unsigned b2 = (unsigned)__builtin_amdgcn_mov_dpp((int)r2, QP, 0xf, 0xf, true);           \
            _Pragma("unroll") for (int t = 0; t < 9; ++t) {      /* (kw, cin) = 9 contiguous bytes of the input row */ \
                const unsigned byte = t < 4 ? (b0 >> (8 * t)) & 0xffu : (t < 8 ? (b1 >> (8 * (t - 4))) & 0xffu : b2); \
                const float v = lut_s[(t % 3) * 256 + byte];     /* same value the fp32 path reads: results are bit-identical */ \
                const float2v* wr = reinterpret_cast<const float2v*>(ws + ((KH) * 9 + t) * 32 + cg);       \
                const float2v vv = {v, v};                                                                 \
                _Pragma("unroll") for (int c = 0; c < 4; ++c) acc2[c] = __builtin_elementwise_fma(vv, wr[c], acc2[c]); \
            }                                                                                              \
        }
        STEM_ROW(0, 0x00)                                     // quad_perm [0,0,0,0]: row 0 from the quad's lane 0
        STEM_ROW(1, 0x55)                                     // [1,1,1,1]
        STEM_ROW(2, 0xaa)                                     // [2,2,2,2]
#undef STEM_ROW
        half8v h, l;
        float vmax = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float v = fmaxf(acc2[c >> 1][c & 1] + bias[cg + c], 0.f);
            vmax = fmaxf(vmax, v);
            h[c] = (_Float16)v;
            l[c] = (_Float16)((v - (float)h[c]) * 2048.f);
        }
        if (live) {
            tise_flag_split_overflow(vmax);
            _Float16* d = out + p * 64 + cg;
            *reinterpret_cast<half8v*>(d) = h;
            *reinterpret_cast<half8v*>(d + 32) = l;
        }
    }
}

// split tensor (N, HW, C), C % 32 == 0 -> fp32 (N, C) mean over the HW positions (AdaptiveAvgPool2d((1,1)) of the
// last block): thread = (image, 8 channels), fixed summation order.
// SPLIT_OUT: the mean is also written as a split row (N, 2C) -- the operand of the classifier layer, which runs as a 1x1
// split-precision convolution on it (trunk.py fc_logits: the last library GEMM of the image loop, round 5).
template <bool SPLIT_OUT>
__global__ __launch_bounds__(256) void split_mean_kernel(const _Float16* __restrict__ x, int N, int HW,
                                                         int C8, float* __restrict__ out, _Float16* __restrict__ out_split) {
    const int C = C8 * 8;
    const int c8 = blockIdx.x * 256 + threadIdx.x;            // one (image = blockIdx.y, 8 channels) element per thread
    if (c8 >= C8) return;
    const int64_t n = blockIdx.y;
    float acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = 0.f;
    const _Float16* q = x + n * HW * (int64_t)C * 2 + tise_ilv_off(c8 * 8, C);
#pragma unroll 4
    for (int s = 0; s < HW; ++s) {
        const half8v vh = *reinterpret_cast<const half8v*>(q + (int64_t)s * C * 2);
        const half8v vl = *reinterpret_cast<const half8v*>(q + (int64_t)s * C * 2 + 32);
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] += (float)vh[i] + (float)vl[i] * (1.f / 2048.f);
    }
    float* d = out + n * (int64_t)C + c8 * 8;
    half8v oh, ol;
    float vmax = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const float v = acc[i] / (float)HW;
        d[i] = v;
        if (SPLIT_OUT) {
            const _Float16 h = (_Float16)v;
            oh[i] = h;
            ol[i] = (_Float16)((v - (float)h) * 2048.f);
            vmax = (fabsf(v) <= vmax) ? vmax : fabsf(v);          // a NaN lands in vmax (the comparison fails) and raises the flag
        }
    }
    if (SPLIT_OUT) {
        _Float16* q2 = out_split + n * (int64_t)C * 2 + tise_ilv_off(c8 * 8, C);
        *reinterpret_cast<half8v*>(q2) = oh;
        *reinterpret_cast<half8v*>(q2 + 32) = ol;
        tise_flag_split_overflow(vmax);
    }
}

inline int grid_for(int64_t total) {
    int64_t b = (total + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (int)b;
}

inline bool aligned4(int64_t a, int b, int c) { return (a % 4 == 0) && (b % 4 == 0) && (c % 4 == 0); }

}  // namespace

extern "C" {

int tise_bias_relu_nhwc(const float* x_dev, int64_t x_ld, int x_off, int64_t pixels, int C, const float* bias_dev,
                        float* out_dev, int64_t out_ld, int out_off, void* stream) {
    if (!x_dev || !bias_dev || !out_dev || pixels < 0 || C <= 0 || !aligned4(x_ld, x_off, C) ||
        !aligned4(out_ld, out_off, C) || x_off + C > x_ld || out_off + C > out_ld)
        return TISE_ERR_INVALID_ARG;
    if (pixels == 0) return TISE_OK;
    hipLaunchKernelGGL(bias_relu_nhwc_kernel, dim3(grid_for(pixels * (C / 4))), dim3(256), 0, (hipStream_t)stream, x_dev,
                       x_ld, x_off, pixels, C / 4, bias_dev, out_dev, out_ld, out_off);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_avgpool3_bias_relu_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                                 const float* bias_dev, float* out_dev, int64_t out_ld, int out_off, void* stream) {
    if (!x_dev || !bias_dev || !out_dev || n < 0 || h <= 0 || w <= 0 || C <= 0 || !aligned4(x_ld, x_off, C) ||
        !aligned4(out_ld, out_off, C) || x_off + C > x_ld || out_off + C > out_ld)
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    hipLaunchKernelGGL(avgpool3_bias_relu_nhwc_kernel, dim3(grid_for((int64_t)n * h * w * (C / 4))), dim3(256), 0,
                       (hipStream_t)stream, x_dev, x_ld, x_off, n, h, w, C / 4, bias_dev, out_dev, out_ld, out_off);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_maxpool3s2_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C, const float* bias_dev,
                         float* out_dev, int64_t out_ld, int out_off, void* stream) {
    if (!x_dev || !out_dev || n < 0 || h < 3 || w < 3 || C <= 0 || !aligned4(x_ld, x_off, C) ||
        !aligned4(out_ld, out_off, C) || x_off + C > x_ld || out_off + C > out_ld)
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    const int g = grid_for((int64_t)n * oh * ow * (C / 4));
    if (bias_dev)
        hipLaunchKernelGGL(maxpool3s2_nhwc_kernel<true>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_dev, x_ld, x_off, n,
                           h, w, C / 4, bias_dev, out_dev, out_ld, out_off);
    else
        hipLaunchKernelGGL(maxpool3s2_nhwc_kernel<false>, dim3(g), dim3(256), 0, (hipStream_t)stream, x_dev, x_ld, x_off,
                           n, h, w, C / 4, bias_dev, out_dev, out_ld, out_off);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_avgpool3_bias_relu_split_nhwc(const float* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                                       const float* bias_dev, void* out_dev, int64_t out_ld, int out_off,
                                       void* stream) {
    if (!x_dev || !bias_dev || !out_dev || n < 0 || h <= 0 || w <= 0 || C <= 0 || C % 8 || x_ld % 4 || x_off % 4 ||
        out_off % 8 || out_ld % 16 || x_off + C > x_ld || out_off + C > out_ld || out_ld > 0x7fffffff ||
        (reinterpret_cast<uintptr_t>(bias_dev) & 15) != 0)
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int64_t cols = (int64_t)n * w * (C / 8);
    static const bool per_output = getenv("TISE_AVGPOOL_PER_OUTPUT") != nullptr;      // A/B switch: round 2's kernel
    if (!per_output && cols < 0x7fffff00LL && x_ld < 0x7fffffffLL && (reinterpret_cast<uintptr_t>(x_dev) & 15) == 0)
        hipLaunchKernelGGL(avgpool3_bias_relu_split_colwalk_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0,
                           (hipStream_t)stream, x_dev, (int)x_ld, x_off, n, h, w, C / 8, bias_dev,
                           reinterpret_cast<_Float16*>(out_dev), (int)out_ld, out_off);
    else
        hipLaunchKernelGGL(avgpool3_bias_relu_split_kernel, dim3(grid_for((int64_t)n * h * w * (C / 8))), dim3(256), 0,
                           (hipStream_t)stream, x_dev, x_ld, x_off, n, h, w, C / 8, bias_dev,
                           reinterpret_cast<_Float16*>(out_dev), (int)out_ld, out_off);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_maxpool3s2_split_nhwc(const void* x_dev, int64_t x_ld, int x_off, int n, int h, int w, int C,
                               void* out_dev, int64_t out_ld, int out_off, void* stream) {
    if (!x_dev || !out_dev || n < 0 || h < 3 || w < 3 || C <= 0 || C % 8 || x_ld % 16 || x_off % 8 || out_ld % 16 ||
        out_off % 8 || x_off + C > x_ld || out_off + C > out_ld || x_ld > 0x7fffffff || out_ld > 0x7fffffff)
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    if (n > 65535 || (int64_t)oh * ow * (C / 8) >= 0x7fffff00LL) return TISE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(maxpool3s2_split_kernel, dim3((unsigned)(((int64_t)oh * ow * (C / 8) + 255) / 256), (unsigned)n), dim3(256), 0,
                       (hipStream_t)stream, reinterpret_cast<const _Float16*>(x_dev), (int)x_ld, x_off, n, h, w, C / 8,
                       reinterpret_cast<_Float16*>(out_dev), (int)out_ld, out_off);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_stem_conv3x3s2_split(const float* x_dev, int n, int h, int w, const float* w_dev, const float* bias_dev,
                              void* out_dev, void* stream) {
    if (!x_dev || !w_dev || !bias_dev || !out_dev || n < 0 || h < 3 || w < 3) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    hipLaunchKernelGGL(stem_conv3x3s2_split_kernel, dim3(grid_for((int64_t)n * oh * ow * 4)), dim3(256), 0,
                       (hipStream_t)stream, x_dev, n, h, w, w_dev, bias_dev, reinterpret_cast<_Float16*>(out_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_stem_conv3x3s2_split_u8(const uint8_t* x_dev, const float* lut_dev, int n, int h, int w, const float* w_dev,
                                 const float* bias_dev, void* out_dev, void* stream) {
    if (!x_dev || !lut_dev || !w_dev || !bias_dev || !out_dev || n < 0 || h < 3 || w < 3) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    const int oh = (h - 3) / 2 + 1, ow = (w - 3) / 2 + 1;
    if ((int64_t)n * oh * ow >= 0x7fffff00LL) return TISE_ERR_UNSUPPORTED;      // 32-bit pixel index in the kernel
    hipLaunchKernelGGL(stem_conv3x3s2_split_u8_kernel, dim3(grid_for((int64_t)n * oh * ow * 4)), dim3(256), 0,
                       (hipStream_t)stream, x_dev, lut_dev, n, h, w, w_dev, bias_dev, reinterpret_cast<_Float16*>(out_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_split_mean_nhwc(const void* x_dev, int n, int hw, int C, float* out_dev, void* stream) {
    if (!x_dev || !out_dev || n < 0 || hw <= 0 || C <= 0 || C % 32) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    if (n > 65535) return TISE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(split_mean_kernel<false>, dim3((unsigned)((C / 8 + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x_dev), n, hw, C / 8, out_dev, (_Float16*)nullptr);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_split_mean_both_nhwc(const void* x_dev, int n, int hw, int C, float* out_dev, void* out_split_dev, void* stream) {
    if (!x_dev || !out_dev || !out_split_dev || n < 0 || hw <= 0 || C <= 0 || C % 32) return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    if (n > 65535) return TISE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(split_mean_kernel<true>, dim3((unsigned)((C / 8 + 255) / 256), (unsigned)n), dim3(256), 0, (hipStream_t)stream,
                       reinterpret_cast<const _Float16*>(x_dev), n, hw, C / 8, out_dev, reinterpret_cast<_Float16*>(out_split_dev));
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // extern "C"

TISE_DEFINE_SPLIT_FLAG_READER(tise_internal_split_flag_trunk_ops)
