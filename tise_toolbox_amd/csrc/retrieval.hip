// Top-1 text retrieval for R-precision (RP-COCO) and the 2-way softmax test of positional alignment (PA).
//
// Replaces, per item, the tail of the reference's loop (text_relevance/RP_coco.py:72-78):
//     logits_per_image = logit_scale * <img / |img|, txt_j / |txt_j|>       (CLIP.forward, third-party `clip`)
//     probs = logits_per_image.softmax(-1);  success = (argmax(probs) == 0)
// and positional_alignment/PA.py:37-42 (probs[0] > 0.6).  The comparisons are made on numbers rounded as CLIP.forward
// and the scripts round them (fp16 on a GPU, fp32 on the CPU path: see rnd<T> below), so near-ties resolve as in
// the reference: first maximum of the ROUNDED softmax output.
//
// The reference encodes the ~100 captions of every item again for every item (30 k items -> 3 M text-tower
// passes, batch 1).  Here every DISTINCT caption is embedded once into a table and an item carries int32 indices
// into it, candidate 0 being the true caption.
//
// One wave per item: the image vector stays in registers (element e = lane + 64 k: coalesced), each candidate
// row is read once (d * 4 or d * 2 bytes, gathered through the index: L2-resident table), dot product and squared
// norm reduced across the wave (fp64 accumulation of the rounded operands), the c logits of the item kept in LDS
// for the softmax pass.
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

// Rounding of CLIP.forward (third-party `clip`, model.py: `logit_scale * image_features @ text_features.t()`, then
// the scripts' `.softmax(dim=-1).cpu().numpy()`): with the fp16 model that clip.load serves on a GPU the scaled
// image features, the logits and the softmax outputs are each rounded to fp16 (fp32 arithmetic inside), with the
// fp32 model of the CPU path to fp32.  The success rules compare THOSE rounded numbers -- np.argmax returns the
// FIRST maximum, so a distractor whose probability rounds to the same half as the true caption's loses (index 0
// wins), and PA's `probs[0] > 0.6` sees the rounded probability.  At logit_scale 100 one fp16 ulp of a logit is
// 0.016-0.03, so this is not a corner case.  rnd<T>() applies the storage rounding of the embeddings' dtype.
template <typename T>
__device__ __forceinline__ float rnd(float v);
template <>
__device__ __forceinline__ float rnd<float>(float v) { return v; }
template <>
__device__ __forceinline__ float rnd<_Float16>(float v) { return (float)(_Float16)v; }

#define RT_MAXC 1024   // candidates per item whose logits are kept in LDS between the two passes

template <typename T, int KMAX>
__global__ __launch_bounds__(256) void cosine_top1_kernel(const T* __restrict__ img, const T* __restrict__ txt,
                                                          const int* __restrict__ index, int64_t n, int c, int d,
                                                          int normalize, float logit_scale, int* __restrict__ top1,
                                                          float* __restrict__ p0) {
    __shared__ float s_logit[4][RT_MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t item = (int64_t)blockIdx.x * 4 + wave;
    if (item >= n) return;
    float a[KMAX];
    double na = 0.0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
        const int e = lane + 64 * k;
        a[k] = e < d ? ld<T>(img + item * d + e) : 0.f;
        na += (double)a[k] * (double)a[k];
    }
    if (normalize) {
        na = wave_sum(na);
        const float nrm = rnd<T>((float)sqrt(na));                             // x.norm(dim=1, keepdim=True): model dtype
#pragma unroll
        for (int k = 0; k < KMAX; ++k) a[k] = rnd<T>(a[k] / nrm);              // x / x.norm()
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) a[k] = rnd<T>(logit_scale * a[k]);          // logit_scale * image_features
    float* lg = s_logit[wave];
    float m = -INFINITY;
    for (int j = 0; j < c; ++j) {
        const int64_t row = index ? (int64_t)index[item * c + j] : item * c + j;
        const T* t = txt + row * d;
        float tv[KMAX];
        double nt = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int e = lane + 64 * k;
            tv[k] = e < d ? ld<T>(t + e) : 0.f;
            nt += (double)tv[k] * (double)tv[k];
        }
        if (normalize) {
            nt = wave_sum(nt);
            const float nrm = rnd<T>((float)sqrt(nt));
#pragma unroll
            for (int k = 0; k < KMAX; ++k) tv[k] = rnd<T>(tv[k] / nrm);
        }
        double dot = 0.0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) dot += (double)a[k] * (double)tv[k];
        const float logit = rnd<T>((float)wave_sum(dot));                       // the matmul's output dtype
        if (lane == 0) lg[j] = logit;
        m = fmaxf(m, logit);
    }
    // softmax over the c stored logits (fp32 arithmetic, output rounded to the dtype), first maximum of the OUTPUT
    double s = 0.0;
    for (int j = lane; j < c; j += 64) s += exp((double)(lg[j] - m));
    s = wave_sum(s);
    const float pmax = rnd<T>((float)(1.0 / s));
    int first = 0x7fffffff;
    for (int j = lane; j < c; j += 64) {
        const float pj = rnd<T>((float)(exp((double)(lg[j] - m)) / s));
        if (pj == pmax && j < first) first = j;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) first = min(first, __shfl_xor(first, off, 64));
    if (lane == 0) {
        top1[item] = first;
        if (p0) p0[item] = rnd<T>((float)(exp((double)(lg[0] - m)) / s));
    }
}

template <typename T>
int launch(const void* img, const void* txt, const int* index, int64_t n, int c, int d, int normalize, float scale,
           int* top1, float* p0, hipStream_t st) {
    const dim3 grid((unsigned)((n + 3) / 4)), block(256);
    const T* a = reinterpret_cast<const T*>(img);
    const T* t = reinterpret_cast<const T*>(txt);
    if (d <= 512) hipLaunchKernelGGL((cosine_top1_kernel<T, 8>), grid, block, 0, st, a, t, index, n, c, d, normalize, scale, top1, p0);
    else hipLaunchKernelGGL((cosine_top1_kernel<T, 16>), grid, block, 0, st, a, t, index, n, c, d, normalize, scale, top1, p0);
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
    if ((n + 3) / 4 > 0x7fffffffLL || c > RT_MAXC) return TISE_ERR_UNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    if (dtype == 0) return launch<float>(img_emb_dev, txt_emb_dev, txt_index_dev, n, c, d, normalize, logit_scale, top1_out_dev, p0_out_dev, st);
    return launch<_Float16>(img_emb_dev, txt_emb_dev, txt_index_dev, n, c, d, normalize, logit_scale, top1_out_dev, p0_out_dev, st);
}
