// Top-1 text retrieval for R-precision (RP-COCO) and the 2-way softmax test of positional alignment (PA).
//
// Replaces, per item, the tail of the reference's loop (text_relevance/RP_coco.py:72-78):
//     logits_per_image = logit_scale * <img / |img|, txt_j / |txt_j|>       (CLIP.forward, third-party `clip`)
//     probs = logits_per_image.softmax(-1);  success = (argmax(probs) == 0)
// and positional_alignment/PA.py:37-42 (probs[0] > 0.6).  softmax is monotone, so top-1 needs only the logits;
// p0 = softmax(logits)[0] is produced as well (PA's threshold test).
//
// The reference encodes the ~100 captions of every item again for every item (30 k items -> 3 M text-tower
// passes, batch 1).  Here every DISTINCT caption is embedded once into a table and an item carries int32 indices
// into it, candidate 0 being the true caption.
//
// One wave per item: the image vector stays in registers (element e = lane + 64 k: coalesced), each candidate
// row is read once (d * 4 or d * 2 bytes, gathered through the index: L2-resident table), dot product and squared
// norm reduced across the wave in fp32 -> fp64 logit, running first-maximum and online log-sum-exp.
// HBM/L2-bound: n * c * d * elem bytes (30 k x 100 x 512 x 2 B = 3.1 GB per image set).
// Fixed evaluation order, no atomics: bitwise reproducible.
#include <hip/hip_fp16.h>
#include "common.h"

namespace {

template <typename T>
__device__ __forceinline__ float ld(const T* p);
template <>
__device__ __forceinline__ float ld<float>(const float* p) { return *p; }
template <>
__device__ __forceinline__ float ld<_Float16>(const _Float16* p) { return (float)*p; }

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void cosine_top1_kernel(const T* __restrict__ img, const T* __restrict__ txt,
                                                          const int* __restrict__ index, int64_t n, int c, int d,
                                                          int normalize, double logit_scale, int* __restrict__ top1,
                                                          float* __restrict__ p0) {
    const int lane = threadIdx.x & 63;
    const int64_t item = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (item >= n) return;
    float a[KMAX];
    double na = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int e = lane + 64 * k;
        a[k] = e < d ? ld<T>(img + item * d + e) : 0.f;
        na += (double)a[k] * (double)a[k];
    }
    na = wave_sum(na);
    const double inv_na = normalize ? 1.0 / sqrt(na) : 1.0;
    int best = 0;
    double best_v = -INFINITY, m = -INFINITY, s = 0.0, l0 = 0.0;
    for (int j = 0; j < c; ++j) {
        const int64_t row = index ? (int64_t)index[item * c + j] : item * c + j;
        const T* t = txt + row * d;
        double dot = 0.0, nt = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            const float v = e < d ? ld<T>(t + e) : 0.f;
            dot += (double)a[k] * (double)v;
            nt += (double)v * (double)v;
        }
        dot = wave_sum(dot);
        nt = wave_sum(nt);
        const double logit = logit_scale * dot * inv_na * (normalize ? 1.0 / sqrt(nt) : 1.0);
        if (j == 0) l0 = logit;
        if (logit > best_v) { best_v = logit; best = j; }   // strict: the first maximum wins, as np.argmax
        if (logit > m) { s = s * exp(m - logit) + 1.0; m = logit; }
        else s += exp(logit - m);
    }
    if (lane == 0) {
        top1[item] = best;
        if (p0) p0[item] = (float)(exp(l0 - m) / s);
    }
}

template <typename T>
int launch(const void* img, const void* txt, const int* index, int64_t n, int c, int d, int normalize, float scale,
           int* top1, float* p0, hipStream_t st) {
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    const T* a = reinterpret_cast<const T*>(img);
    const T* t = reinterpret_cast<const T*>(txt);
    if (d <= 512) hipLaunchKernelGGL((cosine_top1_kernel<T, 8>), grid, block, 0, st, a, t, index, n, c, d, normalize, (double)scale, top1, p0);
    else hipLaunchKernelGGL((cosine_top1_kernel<T, 16>), grid, block, 0, st, a, t, index, n, c, d, normalize, (double)scale, top1, p0);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // namespace

extern "C" int tise_cosine_top1(const void* img_emb_dev, const void* txt_emb_dev, const int32_t* txt_index_dev, int64_t n,
                                int c, int d, int dtype, int normalize, float logit_scale, int32_t* top1_out_dev,
                                float* p0_out_dev, void* stream) {
    if (!img_emb_dev || !txt_emb_dev || !top1_out_dev || n < 0 || c < 1 || d < 1 || d > 1024 || (dtype != 0 && dtype != 1))
        return TISE_ERR_INVALID_ARG;
    if (n == 0) return TISE_OK;
    if ((n + 3) / 4 > 0x7fffffffLL) return TISE_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return launch<float>(img_emb_dev, txt_emb_dev, txt_index_dev, n, c, d, normalize, logit_scale, top1_out_dev, p0_out_dev, st);
    return launch<_Float16>(img_emb_dev, txt_emb_dev, txt_index_dev, n, c, d, normalize, logit_scale, top1_out_dev, p0_out_dev, st);
}
