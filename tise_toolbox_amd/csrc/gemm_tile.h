// 64x64 fp64 output tile on v_mfma_f64_16x16x4_f64 (gfx950), shared by the covariance
// accumulator (fp32 feature rows in) and the Frechet-distance GEMMs (fp64 in).
//
// Workgroup = 256 threads = 4 waves in a 2x2 grid, each wave owns a 32x32 sub-tile
// (2x2 MFMA tiles, 4 x double4 accumulators).  K is walked in steps of 16 through LDS:
// the next K-slab is fetched into registers while the current one feeds the MFMAs.
//
// Operand addressing is by strides so the same tile serves A^T A (feature rows, m
// contiguous), row-major NN and NT products:
//     A(m,k) = A[m*sam + k*sak]      B(k,n) = B[k*sbk + n*sbn]
// The global->LDS fetch picks the lane order that is contiguous in memory for each operand.
//
// MFMA lane maps (cdna_hip_programming.md section 3, f64 form): A[i = lane&15][k = lane>>4],
// B[k = lane>>4][j = lane&15]; D: col = lane&15, row = (lane>>4) + 4*reg.
#pragma once
#include "common.h"

#define GT_BM 64
#define GT_BN 64
#define GT_BK 16
// LDS row pitch in doubles: 80 = 64 + 16 puts the two k-rows a 32-lane half reads with one
// ds_read_b64 on disjoint banks ((a/4) % 64 banking for b64 reads).
#define GT_PITCH 80
#define GT_LDS_DOUBLES (2 * GT_BK * GT_PITCH)

template <typename T>
struct GtFetch {
    double v[4];
    // lim = number of valid entries along the 64-wide (m or n) dimension starting at x0
    __device__ __forceinline__ void load(const T* __restrict__ p, int64_t sx, int64_t sk, int x0, int xlim,
                                         int k0, int klim, int tid) {
        if (sx == 1) {  // contiguous along the 64-wide dimension: 64 consecutive lanes per k row
            const int x = tid & 63;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int k = (tid >> 6) + 4 * q;
                v[q] = (x0 + x < xlim && k0 + k < klim) ? (double)p[(int64_t)(x0 + x) * sx + (int64_t)(k0 + k) * sk] : 0.0;
            }
        } else {        // contiguous along k: 16 consecutive lanes per row
            const int k = tid & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int x = (tid >> 4) + 16 * q;
                v[q] = (x0 + x < xlim && k0 + k < klim) ? (double)p[(int64_t)(x0 + x) * sx + (int64_t)(k0 + k) * sk] : 0.0;
            }
        }
    }
    __device__ __forceinline__ void store(double* __restrict__ lds, int64_t sx, int tid) const {
        if (sx == 1) {
            const int x = tid & 63;
#pragma unroll
            for (int q = 0; q < 4; ++q) lds[((tid >> 6) + 4 * q) * GT_PITCH + x] = v[q];
        } else {
            const int k = tid & 15;
#pragma unroll
            for (int q = 0; q < 4; ++q) lds[k * GT_PITCH + (tid >> 4) + 16 * q] = v[q];
        }
    }
};

// acc[tm][tn] += A(m0.., :) * B(:, n0..)  over k in [0, K)
template <typename TA, typename TB>
__device__ __forceinline__ void gemm_tile_64x64(const TA* __restrict__ A, int64_t sam, int64_t sak,
                                                const TB* __restrict__ B, int64_t sbk, int64_t sbn, int M, int N,
                                                int K, int m0, int n0, double4_t (&acc)[2][2], double* lds) {
    double* As = lds;
    double* Bs = lds + GT_BK * GT_PITCH;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fi = lane & 15, fk = lane >> 4;

    GtFetch<TA> fa;
    GtFetch<TB> fb;
    fa.load(A, sam, sak, m0, M, 0, K, tid);
    fb.load(B, sbn, sbk, n0, N, 0, K, tid);
    for (int k0 = 0; k0 < K; k0 += GT_BK) {
        fa.store(As, sam, tid);
        fb.store(Bs, sbn, tid);
        __syncthreads();
        if (k0 + GT_BK < K) {
            fa.load(A, sam, sak, m0, M, k0 + GT_BK, K, tid);
            fb.load(B, sbn, sbk, n0, N, k0 + GT_BK, K, tid);
        }
#pragma unroll
        for (int kk = 0; kk < GT_BK; kk += 4) {
            const double a0 = As[(kk + fk) * GT_PITCH + wr * 32 + fi];
            const double a1 = As[(kk + fk) * GT_PITCH + wr * 32 + 16 + fi];
            const double b0 = Bs[(kk + fk) * GT_PITCH + wc * 32 + fi];
            const double b1 = Bs[(kk + fk) * GT_PITCH + wc * 32 + 16 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
}

// write / accumulate the wave's 32x32 sub-tile.  MODE 0: C = acc, 1 (true): C += acc, 2: C -= acc.
template <int MODE>
__device__ __forceinline__ void gemm_tile_store(double* __restrict__ C, int64_t ldc, int M, int N, int m0, int n0,
                                                const double4_t (&acc)[2][2]) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wr * 32 + tm * 16 + (lane >> 4) + 4 * r;
                const int col = n0 + wc * 32 + tn * 16 + (lane & 15);
                if (row < M && col < N) {
                    double* p = C + (int64_t)row * ldc + col;
                    if (MODE == 1) *p += acc[tm][tn][r];
                    else if (MODE == 2) *p -= acc[tm][tn][r];
                    else *p = acc[tm][tn][r];
                }
            }
}
