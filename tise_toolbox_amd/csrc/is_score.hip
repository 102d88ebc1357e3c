// IS* reduction: softmax-with-temperature + per-split KL/entropy sums, one pass, fp64 accumulation.
//
// Replaces tf.div(logits, T) + tf.nn.softmax (reference image_realism/IS/coco/inception_score_star_coco.py:107-108)
// and the 10-split loop  kl = part*(log part - log mean(part,0)); exp(mean(sum(kl,1)))   (:52-60; same loop in
// IS/bird/inception_score_star_bird.py:97-108 with tf.slice(...,[0,1]) :189; per-row scipy entropy form in
// object_fidelity/O-IS/object_centric_inception_score.py:69-81).
//
// Identity used:  mean_i sum_c p_ic (log p_ic - log pbar_c)
//                                = (1/n) sum_i sum_c p_ic log p_ic  -  sum_c pbar_c log pbar_c
// so per split k only A_k = sum_i sum_c p_ic log p_ic and B_kc = sum_i p_ic are kept; both are
// additive over batches and GPUs.  log p_ic = z_ic - lse_i with z = logit / T, hence
// sum_c p_ic log p_ic = (sum_c p_ic z_ic) - lse_i.
//
// Kernels (all HBM-bound: logits are read twice, 2 * 4 * C bytes per row)
//   is_row_kernel   one wave per row: row max, log-sum-exp, a_i = sum_c p z - lse   (wave shuffles)
//   is_col_kernel   32 columns x 32 row phases per workgroup, fixed row order: B_kc += exp(z_ic - lse_i); one extra
//                   block folds a_i into A_k.  Fixed order => bitwise reproducible, no atomics.
//   is_finalize_kernel  scores, mean, std (ddof 0).
#include "common.h"

namespace {

// local row range [r0, r1) of this call (rows idx_base .. idx_base+rows-1) that falls in split k
__device__ __forceinline__ void split_range(int k, int64_t idx_base, int64_t rows, int64_t n_total, int splits,
                                            int rule, int64_t* r0, int64_t* r1) {
    int64_t lo, hi;
    if (rule == 0) {            // coco / bird: [k*N/splits, (k+1)*N/splits)
        lo = ((int64_t)k * n_total) / splits;
        hi = ((int64_t)(k + 1) * n_total) / splits;
    } else {                    // O-IS: [k*(N/splits), (k+1)*(N/splits)), tail dropped
        const int64_t per = n_total / splits;
        lo = (int64_t)k * per;
        hi = lo + per;
    }
    lo -= idx_base;
    hi -= idx_base;
    *r0 = lo < 0 ? 0 : lo;
    *r1 = hi > rows ? rows : hi;
}

__device__ __forceinline__ int64_t split_size(int k, int64_t n_total, int splits, int rule) {
    if (rule == 0) return ((int64_t)(k + 1) * n_total) / splits - ((int64_t)k * n_total) / splits;
    return n_total / splits;
}

__global__ __launch_bounds__(256) void is_row_kernel(const float* __restrict__ logits, int64_t rows, int64_t ld, int C,
                                                     int c0, double inv_t, double* __restrict__ lse_out,
                                                     double* __restrict__ a_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* z = logits + row * ld + c0;
    double m = -INFINITY;
    for (int c = lane; c < C; c += 64) m = fmax(m, (double)z[c] * inv_t);
    m = wave_max(m);
    double se = 0.0, sz = 0.0;
    for (int c = lane; c < C; c += 64) {
        const double zz = (double)z[c] * inv_t - m;
        const double e = exp(zz);
        se += e;
        sz += e * zz;
    }
    se = wave_sum(se);
    sz = wave_sum(sz);
    if (lane == 0) {
        const double l = log(se);
        lse_out[row] = m + l;                 // log sum exp of z
        a_out[row] = sz / se - l;             // sum_c p (z - m) - log(se) = sum_c p log p
    }
}

// blocks [0, nblk_c): 1024 threads = 32 columns x 32 row phases, rows in fixed order (the first version ran 64 columns
// x 4 phases in 16 workgroups: every thread walked 250 rows of fp64 exp by itself, 83-97 us per 1000 x 1000 batch).
// block nblk_c: folds a_i into A_k.
#define ISC_COLS 32
#define ISC_PH 32
__global__ __launch_bounds__(1024) void is_col_kernel(const float* __restrict__ logits, int64_t rows, int64_t ld, int C,
                                                     int c0, double inv_t, const double* __restrict__ lse,
                                                     const double* __restrict__ a, int64_t idx_base, int64_t n_total,
                                                     int splits, int rule, int nblk_c, double* __restrict__ acc) {
    __shared__ double part[ISC_PH][ISC_COLS + 1];
    double* A = acc;
    double* B = acc + splits;
    const int ph = threadIdx.x / ISC_COLS, lc = threadIdx.x % ISC_COLS;
    if ((int)blockIdx.x == nblk_c) {
        // A_k += sum of a_i over the rows of split k present in this call (fixed order)
        for (int k = 0; k < splits; ++k) {
            int64_t r0, r1;
            split_range(k, idx_base, rows, n_total, splits, rule, &r0, &r1);
            if (r1 <= r0) continue;                      // uniform across the block
            double v = 0.0;
            for (int64_t r = r0 + threadIdx.x; r < r1; r += 1024) v += a[r];
            v = wave_sum(v);
            if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][0] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = 0.0;
                for (int w = 0; w < 16; ++w) t += part[w][0];
                A[k] += t;
            }
            __syncthreads();
        }
        return;
    }
    const int c = blockIdx.x * ISC_COLS + lc;
    for (int k = 0; k < splits; ++k) {
        int64_t r0, r1;
        split_range(k, idx_base, rows, n_total, splits, rule, &r0, &r1);
        if (r1 <= r0) continue;                          // uniform across the block
        double sum = 0.0;
        if (c < C)
            for (int64_t r = r0 + ph; r < r1; r += ISC_PH) sum += exp((double)logits[r * ld + c0 + c] * inv_t - lse[r]);
        part[ph][lc] = sum;
        __syncthreads();
        if (ph == 0 && c < C) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < ISC_PH; ++q) t += part[q][lc];
            B[(int64_t)k * C + c] += t;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void is_finalize_kernel(const double* __restrict__ acc, int C, int64_t n_total,
                                                          int splits, int rule, double* __restrict__ out) {
    __shared__ double part[4];
    __shared__ double scores[64];
    const double* A = acc;
    const double* B = acc + splits;
    for (int k = 0; k < splits; ++k) {
        const double nk = (double)split_size(k, n_total, splits, rule);
        double h = 0.0;
        for (int c = threadIdx.x; c < C; c += 256) {
            const double pbar = B[(int64_t)k * C + c] / nk;
            if (pbar > 0.0) h += pbar * log(pbar);
        }
        h = wave_sum(h);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = h;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double hh = ((part[0] + part[1]) + part[2]) + part[3];
            const double s = exp(A[k] / nk - hh);
            if (k < 64) scores[k] = s;
            out[2 + k] = s;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double mean = 0.0;
        for (int k = 0; k < splits; ++k) mean += out[2 + k];
        mean /= splits;
        double var = 0.0;
        for (int k = 0; k < splits; ++k) { const double dlt = out[2 + k] - mean; var += dlt * dlt; }
        out[0] = mean;
        out[1] = sqrt(var / splits);
    }
}

}  // namespace

extern "C" {

int tise_is_update(const float* logits_dev, int64_t rows, int64_t ld, int C, double temperature, int drop_first,
                   int64_t idx_base, int64_t n_total, int splits, int split_rule, double* acc_dev, double* ws_dev,
                   void* stream) {
    if (rows < 0 || C <= (drop_first ? 1 : 0) || ld < C || !(temperature > 0.0) || splits <= 0 || n_total <= 0 ||
        idx_base < 0 || idx_base + rows > n_total || (split_rule != 0 && split_rule != 1) || !acc_dev || !ws_dev ||
        (rows > 0 && !logits_dev))
        return TISE_ERR_INVALID_ARG;
    if (rows == 0) return TISE_OK;
    hipStream_t st = (hipStream_t)stream;
    const int c0 = drop_first ? 1 : 0;
    const int Ce = C - c0;
    const double inv_t = 1.0 / temperature;
    double* lse = ws_dev;
    double* a = ws_dev + rows;
    hipLaunchKernelGGL(is_row_kernel, dim3((unsigned)ceil_div64(rows, 4)), dim3(256), 0, st, logits_dev, rows, ld, Ce, c0,
                       inv_t, lse, a);
    TISE_LAUNCH_CHECK();
    const int nblk_c = ceil_div(Ce, ISC_COLS);
    hipLaunchKernelGGL(is_col_kernel, dim3(nblk_c + 1), dim3(1024), 0, st, logits_dev, rows, ld, Ce, c0, inv_t, lse, a,
                       idx_base, n_total, splits, split_rule, nblk_c, acc_dev);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_is_finalize(const double* acc_dev, int C_eff, int64_t n_total, int splits, int split_rule, double* out_dev,
                     void* stream) {
    if (!acc_dev || !out_dev || C_eff <= 0 || splits <= 0 || n_total <= 0 || (split_rule != 0 && split_rule != 1))
        return TISE_ERR_INVALID_ARG;
    hipLaunchKernelGGL(is_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc_dev, C_eff, n_total, splits,
                       split_rule, out_dev);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // extern "C"
