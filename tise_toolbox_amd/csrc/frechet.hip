// Frechet distance between two Gaussians, all fp64, all on device.
//
// Replaces calculate_frechet_distance (reference image_realism/FID/fid_score.py:121-171), whose cost
// is scipy.linalg.sqrtm(sigma1.dot(sigma2)) -- a complex Schur decomposition of a non-symmetric
// product on the host.  Only the TRACE of that square root is used (:169), and
//
//     Tr sqrtm(S1 S2) = sum_i sqrt(lambda_i(S1 S2)) = sum_i sqrt(lambda_i(L^T S2 L)),   S1 = L L^T,
//
// because S1 S2 = L (L^T S2) and L^T S2 L share their non-zero eigenvalues.  L^T S2 L is symmetric
// positive semi-definite, so a symmetric eigenvalue solver suffices (north_star: "symmetric
// eigendecomposition for the FID matrix square root").  Pipeline:
//
//   1. pchol_*      L^T by diagonally pivoted Cholesky (no row swaps: the permutation only orders the
//                   columns of L, and L L^T = S1 holds in the original row order).  Rank revealing:
//                   runs until the largest remaining pivot is <= 0, so a covariance estimated from
//                   N < d samples (BASELINE config 1: N = 1000, d = 2048) simply yields r < d columns.
//                   Left-looking: column k costs one (k x d) matrix-vector product.     HBM/L2 bound.
//   2. gemm_f64     T1^T = L^T S2 (r x d), M = L^T T1 (r x r)  on v_mfma_f64_16x16x4_f64.   MFMA bound.
//   3. sytrd_*      Householder tridiagonalisation of M; the rank-2 update of step k-1 is fused with
//                   the matrix-vector product of step k: one read-modify-write pass over the trailing
//                   block per column.                                                   HBM/L2 bound.
//   4. bisect       all r eigenvalues of the tridiagonal matrix by Sturm-count bisection, one thread
//                   per eigenvalue (embarrassingly parallel, latency bound).
//   5. finish       fid = |mu1-mu2|^2 + tr S1 + tr S2 - 2 sum sqrt(max(lambda, 0)).
//
// Steps 1 and 3 are sequences of 2 launches per column (a wide kernel + a single-workgroup kernel);
// launch boundaries (~1.5 us, MI355X_MICROARCH "boundary") are cheaper than software grid barriers
// (~4 us) on this part, so they are not fused into a persistent kernel.
//
// The reference's two rescue branches (add eps to both diagonals when sqrtm returns non-finite
// values, :156-160; ValueError on a large imaginary residue, :163-167) cannot trigger in this
// formulation: non-finite input is reported through flag bit 0 and the Python mirror applies the
// reference's eps retry through `diag_offset`.
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include "common.h"
#include "gemm_tile.h"

#define PCHOL_NP 8          // k-chunks of the left-looking matvec
#define SY_RC 256           // rows per row-chunk of the fused sytrd update/matvec (4 phases x 64 rows)
#define SY_CC 64            // columns per workgroup
#define SYF_G 512           // workgroups of the fused one-launch-per-column tridiagonalisation
#define SYF_MAX_N 6144      // 3 vectors of n doubles in LDS (144 KB)

struct FrState {
    int piv;        // next pivot (pchol)
    int rank;       // numerical rank found so far
    int done;       // pchol finished
    int nonfinite;  // a non-finite value was met
    double tau;     // tau of the current Householder reflector
    double tau_prev;
    double glo, ghi; // Gershgorin interval
    double pivmin;
};

struct tise_frechet {
    int d;
    double *lt, *t1, *m;      // d*d each
    double *partial;          // max(PCHOL_NP, d/SY_RC + 1) * d
    double *diag, *colbuf;    // d each
    double *va, *vb, *w;      // d each
    double *td, *te, *eig;    // d each
    double* fz;               // fused tridiagonalisation: p[2][d] | partial dots [2][SYF_G] | tau[d]
    int* chosen;              // d
    FrState* st;
    int profiling;            // record phase events (tise_frechet_set_profiling)
    hipEvent_t ev[6];         // start | pchol end | gemm end | sytrd end | bisect end | finish end
    hipEvent_t evp[2];        // tise_frechet_prefactor: start | end (on the stream it ran on)
    int prefactored;          // h->lt holds the factor of the matrix given to tise_frechet_prefactor
    int lt_tri;               // h->lt came from the unpivoted factorisation: LT[a][i] = 0 for i < 64 * (a / 64)
    int pf_rank;
    int last_rank;
};

namespace {

// ------------------------------------------------------------------ block argmax helper (1024 threads)
__device__ __forceinline__ void block_argmax(double v, int idx, double* s_val, int* s_idx, double* out_v, int* out_i) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const double ov = __shfl_xor(v, off, 64);
        const int oi = __shfl_xor(idx, off, 64);
        if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
    }
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_val[wave] = v; s_idx[wave] = idx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double bv = s_val[0]; int bi = s_idx[0];
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i)
            if (s_val[i] > bv || (s_val[i] == bv && s_idx[i] < bi)) { bv = s_val[i]; bi = s_idx[i]; }
        s_val[0] = bv; s_idx[0] = bi;
    }
    __syncthreads();
    *out_v = s_val[0];
    *out_i = s_idx[0];
    __syncthreads();
}

__device__ __forceinline__ double block_sum(double v, double* s_val) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) s_val[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) t += s_val[i];
        s_val[0] = t;
    }
    __syncthreads();
    const double r = s_val[0];
    __syncthreads();
    return r;
}

// ------------------------------------------------------------------ pivoted Cholesky
__global__ __launch_bounds__(1024) void pchol_init_kernel(const double* __restrict__ S, int d, double off,
                                                          double* __restrict__ diag, int* __restrict__ chosen,
                                                          FrState* __restrict__ st) {
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    double bv = -INFINITY; int bi = 0x7fffffff; int bad = 0;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const double v = S[(int64_t)i * d + i] + off;
        diag[i] = v;
        chosen[i] = 0;
        if (!isfinite(v)) bad = 1;
        if (v > bv) { bv = v; bi = i; }
    }
    bad = __syncthreads_or(bad);
    double v; int idx;
    block_argmax(bv, bi, s_val, s_idx, &v, &idx);
    if (threadIdx.x == 0) {
        st->piv = idx < d ? idx : 0;
        st->rank = 0;
        st->done = 0;
        st->nonfinite = bad;
    }
}

// partial[c][i] = sum_{j in chunk c of [0,k)} LT[j][i] * LT[j][p]
__global__ __launch_bounds__(256) void pchol_partial_kernel(const double* __restrict__ LT, int d, int k,
                                                            const FrState* __restrict__ st,
                                                            double* __restrict__ partial) {
    if (st->done) return;
    __shared__ double part[4][64];
    const int p = st->piv;
    const int lc = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int i = blockIdx.x * 64 + lc;
    const int jlen = (k + PCHOL_NP - 1) / PCHOL_NP;
    const int j0 = blockIdx.y * jlen;
    const int j1 = min(k, j0 + jlen);
    double acc = 0.0;
    if (i < d)
        for (int j = j0 + ph; j < j1; j += 4) acc += LT[(int64_t)j * d + i] * LT[(int64_t)j * d + p];
    part[ph][lc] = acc;
    __syncthreads();
    if (ph == 0 && i < d) partial[(int64_t)blockIdx.y * d + i] = ((part[0][lc] + part[1][lc]) + part[2][lc]) + part[3][lc];
}

__global__ __launch_bounds__(1024) void pchol_finish_kernel(const double* __restrict__ S, int d, int k, double off,
                                                            const double* __restrict__ partial,
                                                            double* __restrict__ LT, double* __restrict__ diag,
                                                            double* __restrict__ colbuf, int* __restrict__ chosen,
                                                            FrState* __restrict__ st) {
    if (st->done) return;
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    __shared__ double s_piv;
    const int p = st->piv;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        double c = S[(int64_t)p * d + i] + (i == p ? off : 0.0);   // row p of the symmetric input = column p
        double sub = 0.0;
#pragma unroll
        for (int q = 0; q < PCHOL_NP; ++q) sub += partial[(int64_t)q * d + i];
        c -= sub;
        colbuf[i] = c;
        if (i == p) s_piv = c;
    }
    __syncthreads();
    const double piv = s_piv;
    if (!(piv > 0.0) || !isfinite(piv)) {          // numerical rank reached (or NaN): stop
        if (threadIdx.x == 0) {
            st->done = 1;
            if (!isfinite(piv)) st->nonfinite = 1;
        }
        return;
    }
    const double sq = sqrt(piv);
    double bv = -INFINITY; int bi = 0x7fffffff;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        double l;
        if (i == p) l = sq;
        else if (chosen[i]) l = 0.0;
        else l = colbuf[i] / sq;
        LT[(int64_t)k * d + i] = l;
        const double dv = diag[i] - l * l;
        diag[i] = dv;
        if (i != p && !chosen[i] && dv > bv) { bv = dv; bi = i; }
    }
    double v; int idx;
    block_argmax(bv, bi, s_val, s_idx, &v, &idx);
    if (threadIdx.x == 0) {
        chosen[p] = 1;
        st->rank = k + 1;
        if (k + 1 >= d || idx >= d) st->done = 1;
        else st->piv = idx;
    }
}

// ------------------------------------------------------------------ blocked pivoted Cholesky (d <= 2048)
// The column-at-a-time version above costs two launches and a (k x d) matrix-vector product per pivot
// (measured 41 ms at d = 2048: 2 x 2048 launches of ~8-9 us).  Blocked form (LAPACK dpstrf idea): pivots are
// processed PCB_NB at a time by ONE 1024-thread workgroup that keeps the block's columns of L in registers
// (row i -> thread i % 1024; 2 rows x 16 columns = 32 doubles per thread at d = 2048; 1024 threads cap a thread at 128 VGPRs), so a pivot step needs
// only a 32-term dot product per row plus one block-wide argmax; the contribution of finished blocks is
// folded into a working copy of the matrix by a rank-16 update on the fp64 MFMA tile (one launch per block).
#define PCB_NB 16

__global__ __launch_bounds__(256) void pchol_copy_kernel(const double* __restrict__ S, int d, double off,
                                                         double* __restrict__ work) {
    const int64_t total = (int64_t)d * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / d), j = (int)(e - (int64_t)i * d);
        work[e] = S[e] + (i == j ? off : 0.0);
    }
}

template <int RPT>
__global__ __launch_bounds__(1024) void pchol_panel_kernel(const double* __restrict__ work, int d, int k0,
                                                           double* __restrict__ LT, double* __restrict__ diag,
                                                           int* __restrict__ chosen, FrState* __restrict__ st) {
    if (st->done) return;
    __shared__ double s_val[16];
    __shared__ int s_idx[16];
    __shared__ double s_lp[PCB_NB];
    __shared__ double s_piv;
    const int tid = threadIdx.x;
    double lreg[RPT][PCB_NB];
    double dg[RPT];
    int ch[RPT];
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = tid + 1024 * r;
        dg[r] = i < d ? diag[i] : -INFINITY;
        ch[r] = i < d ? chosen[i] : 1;
#pragma unroll
        for (int c = 0; c < PCB_NB; ++c) lreg[r][c] = 0.0;
    }
    int p = st->piv;
    int done_at = -1;
    bool nonfinite = false;
    for (int j = 0; j < PCB_NB; ++j) {
        const int k = k0 + j;
        if (k >= d) break;
        // the pivot row's coefficients inside this block, broadcast through LDS
#pragma unroll
        for (int r = 0; r < RPT; ++r)
            if (tid + 1024 * r == p) {
#pragma unroll
                for (int c = 0; c < PCB_NB; ++c) s_lp[c] = lreg[r][c];
            }
        __syncthreads();
        double colv[RPT];
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = tid + 1024 * r;
            double acc = i < d ? work[(int64_t)p * d + i] : 0.0;      // row p of the symmetric working matrix
#pragma unroll
            for (int c = 0; c < PCB_NB; ++c)
                if (c < j) acc -= lreg[r][c] * s_lp[c];
            colv[r] = acc;
            if (i == p) s_piv = acc;
        }
        __syncthreads();
        const double piv = s_piv;
        if (!(piv > 0.0) || !isfinite(piv)) {                         // numerical rank reached (uniform)
            done_at = k;
            nonfinite = !isfinite(piv);
            break;
        }
        const double sq = sqrt(piv);
        double bv = -INFINITY;
        int bi = 0x7fffffff;
#pragma unroll
        for (int r = 0; r < RPT; ++r) {
            const int i = tid + 1024 * r;
            if (i < d) {
                double l;
                if (i == p) { l = sq; ch[r] = 1; }
                else if (ch[r]) l = 0.0;
                else l = colv[r] / sq;
#pragma unroll
                for (int c = 0; c < PCB_NB; ++c)
                    if (c == j) lreg[r][c] = l;
                LT[(int64_t)k * d + i] = l;
                dg[r] -= l * l;
                if (!ch[r] && dg[r] > bv) { bv = dg[r]; bi = i; }
            }
        }
        double v;
        int idx;
        block_argmax(bv, bi, s_val, s_idx, &v, &idx);
        if (k + 1 >= d || idx >= d) { done_at = k + 1; break; }
        p = idx;
    }
#pragma unroll
    for (int r = 0; r < RPT; ++r) {
        const int i = tid + 1024 * r;
        if (i < d) { diag[i] = dg[r]; chosen[i] = ch[r]; }
    }
    if (tid == 0) {
        if (done_at >= 0) { st->done = 1; st->rank = done_at; if (nonfinite) st->nonfinite = 1; }
        else { st->rank = min(k0 + PCB_NB, d); st->piv = p; }
    }
}

// work -= L_blk L_blk^T for the block of columns [k0, k0 + nb) (rows of LT), every 64x64 tile
__global__ __launch_bounds__(256) void pchol_trail_kernel(const double* __restrict__ LT, int d, int k0,
                                                          const FrState* __restrict__ st, double* __restrict__ work) {
    if (st->done) return;
    __shared__ double lds[GT_LDS_DOUBLES];
    const int nb = min(PCB_NB, d - k0);
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double* Lb = LT + (int64_t)k0 * d;
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    gemm_tile_64x64<double, double>(Lb, 1, d, Lb, d, 1, d, d, nb, m0, n0, acc, lds);
    gemm_tile_store<2>(work, d, d, d, m0, n0, acc);
}

// ------------------------------------------------------------------ unpivoted blocked Cholesky (fast path, d <= 2048)
// Pivoting is there to reveal the rank when N < d; a covariance of full numerical rank does not need it, and any
// factor with L L^T = S gives the same eigenvalues of L^T S2 L.  So the factor is first attempted in natural order,
// right-looking, 64 columns at a time -- one launch factors the diagonal block (every workgroup for itself, in
// LDS) and solves its 64 rows of the panel against it, one launch applies the panel to the trailing lower triangle
// on the fp64 MFMA tile: 2 x 32 launches at d = 2048 instead of 128 sixteen-pivot panels with a block-wide argmax
// per pivot (12.6 ms).  A pivot that is not safely positive (<= 1e-12 of the largest diagonal entry, or non-finite)
// raises st->nonpd and the caller starts over with the pivoted, rank-revealing factorisation.
#define CHB 64

__global__ __launch_bounds__(256) void chol_panel_kernel(const double* __restrict__ work, int d, int k0, double tol,
                                                         double* __restrict__ LT, int* __restrict__ fail) {
    __shared__ double D[CHB][CHB + 1];                            // the diagonal block, then its factor L11 (lower)
    __shared__ double X[CHB][CHB + 1];                            // this workgroup's 64 rows of the panel
    __shared__ int s_bad;
    const int tid = threadIdx.x;
    const int nb = min(CHB, d - k0);
    if (tid == 0) s_bad = 0;
    for (int e = tid; e < CHB * CHB; e += 256) {
        const int r = e / CHB, c = e - r * CHB;
        D[r][c] = (r < nb && c < nb && c <= r) ? work[(int64_t)(k0 + r) * d + (k0 + c)] : (r == c ? 1.0 : 0.0);
    }
    __syncthreads();
    // ---- potrf of the 64 x 64 block in LDS (right-looking; thread (r, q): row r = tid / 4, columns q, q+4, ...) ----
    const int pr = tid >> 2, pq = tid & 3;
    for (int j = 0; j < nb; ++j) {
        const double piv = D[j][j];
        if (!(piv > tol) || !isfinite(piv)) {                     // uniform: every thread reads the same value
            if (tid == 0) s_bad = 1;
            break;
        }
        const double inv = 1.0 / sqrt(piv);
        __syncthreads();
        if (tid < CHB && tid >= j && tid < nb) D[tid][j] = (tid == j) ? sqrt(piv) : D[tid][j] * inv;
        __syncthreads();
        if (pr > j && pr < nb) {
            const double lrj = D[pr][j];
            for (int c = j + 1 + pq; c <= pr; c += 4) D[pr][c] -= lrj * D[c][j];
        }
        __syncthreads();
    }
    __syncthreads();
    if (s_bad) {
        if (tid == 0) atomicOr(fail, 1);
        return;
    }
    const int row0 = k0 + CHB * (int)blockIdx.x;                  // first matrix row of this workgroup
    if (blockIdx.x == 0) {
        // the diagonal block itself: LT[k0 + c][k0 + r] = L11[r][c] (zero above the diagonal)
        for (int e = tid; e < CHB * CHB; e += 256) {
            const int c = e / CHB, r = e - c * CHB;               // r fastest: contiguous in LT's row k0 + c
            if (r < nb && c < nb) LT[(int64_t)(k0 + c) * d + (k0 + r)] = (c <= r) ? D[r][c] : 0.0;
        }
        return;
    }
    // ---- rows row0 .. row0+63 of the panel: X = work[rows][k0 .. k0+63], solve X L11^T = A21 row by row -------------
    const int nrows = min(CHB, d - row0);
    for (int e = tid; e < CHB * CHB; e += 256) {
        const int r = e / CHB, c = e - r * CHB;
        X[r][c] = (r < nrows && c < nb) ? work[(int64_t)(row0 + r) * d + (k0 + c)] : 0.0;
    }
    __syncthreads();
    // thread (r, q): row r, partial dots over columns q, q+4, ... (the four lanes of a row are neighbours in a wave)
    for (int j = 0; j < nb; ++j) {
        double acc = 0.0;
        for (int c = pq; c < j; c += 4) acc += X[pr][c] * D[j][c];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (pq == 0) X[pr][j] = (X[pr][j] - acc) / D[j][j];
        __syncthreads();                                          // column j of every row is final before column j+1 reads it
    }
    for (int e = tid; e < CHB * CHB; e += 256) {
        const int c = e / CHB, r = e - c * CHB;                   // r fastest: contiguous in LT's row k0 + c
        if (r < nrows && c < nb) LT[(int64_t)(k0 + c) * d + (row0 + r)] = X[r][c];
    }
}

// work -= L_panel L_panel^T on the trailing LOWER triangle (64 x 64 tiles with row tile >= column tile)
__global__ __launch_bounds__(256) void chol_trail_kernel(const double* __restrict__ LT, int d, int k0, const int* __restrict__ fail,
                                                         double* __restrict__ work) {
    if (blockIdx.x > blockIdx.y || *fail) return;
    __shared__ double lds[GT_LDS_DOUBLES];
    const int nb = min(CHB, d - k0);
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const double* Lb = LT + (int64_t)k0 * d;
    const int t0 = k0 + CHB;
    const int m0 = t0 + blockIdx.y * GT_BM, n0 = t0 + blockIdx.x * GT_BN;
    gemm_tile_64x64<double, double>(Lb, 1, d, Lb, d, 1, d, d, nb, m0, n0, acc, lds);
    gemm_tile_store<2>(work, d, d, d, m0, n0, acc);
}

// rows of LT above the diagonal block of their panel are zero (L is lower triangular): LT[k][i] = 0 for i < 64 * (k / 64)
__global__ void chol_zero_upper_kernel(double* __restrict__ LT, int d) {
    const int64_t total = (int64_t)d * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int k = (int)(e / d), i = (int)(e - (int64_t)k * d);
        if (i < (k / CHB) * CHB) LT[e] = 0.0;
    }
}

__global__ void chol_finish_kernel(const int* __restrict__ fail, int d, FrState* __restrict__ st) {
    if (threadIdx.x == 0 && blockIdx.x == 0) { st->done = 1; st->rank = *fail ? 0 : d; st->nonfinite = 0; }
}

__global__ __launch_bounds__(1024) void chol_maxdiag_kernel(const double* __restrict__ work, int d, double* __restrict__ out, int* __restrict__ fail) {
    __shared__ double s_val[16];
    double m = 0.0;
    bool bad = false;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const double v = work[(int64_t)i * d + i];
        if (!isfinite(v)) bad = true;
        m = fmax(m, v);
    }
    m = wave_max(m);
    if ((threadIdx.x & 63) == 0) s_val[threadIdx.x >> 6] = m;
    const int anybad = __syncthreads_or(bad ? 1 : 0);
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmax(m, s_val[i]);
        out[0] = m;
        *fail = anybad ? 1 : 0;
    }
}

// zero rows [r, d) of LT so later consumers may ignore the rank
__global__ void zero_rows_kernel(double* __restrict__ LT, int d, int r0) {
    const int64_t total = (int64_t)(d - r0) * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x)
        LT[(int64_t)r0 * d + e] = 0.0;
}

// ------------------------------------------------------------------ GEMM (fp64 MFMA)
__global__ __launch_bounds__(256) void gemm_f64_kernel(const double* __restrict__ A, int64_t sam, int64_t sak,
                                                       const double* __restrict__ B, int64_t sbk, int64_t sbn,
                                                       double* __restrict__ C, int64_t ldc, int M, int N, int K) {
    __shared__ double lds[GT_LDS_DOUBLES];
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    gemm_tile_64x64<double, double>(A, sam, sak, B, sbk, sbn, M, N, K, m0, n0, acc, lds);
    gemm_tile_store<0>(C, ldc, M, N, m0, n0, acc);
}

// The two GEMMs of the distance when L^T is block upper triangular (unpivoted factor: LT[a][i] = 0 for i < 64 (a / 64)):
// the contraction of output row tile m0 starts at k = m0 (half the flops of T1^T = L^T S2), and of M = L^T T1 only the
// tiles on and above the diagonal are computed (M is symmetric; mirror_upper_kernel fills the rest): a third of its
// flops.  UPPER = 1: skip tiles with n0 < m0.
template <int UPPER>
__global__ __launch_bounds__(256) void gemm_f64_tri_kernel(const double* __restrict__ A, int64_t sam, int64_t sak,
                                                           const double* __restrict__ B, int64_t sbk, int64_t sbn,
                                                           double* __restrict__ C, int64_t ldc, int M, int N, int K) {
    const int m0 = blockIdx.y * GT_BM, n0 = blockIdx.x * GT_BN;
    if (UPPER && n0 < m0) return;
    __shared__ double lds[GT_LDS_DOUBLES];
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    const int k0 = m0 < K ? m0 : K;                            // rows m0.. of L^T are zero in columns < m0
    gemm_tile_64x64<double, double>(A + (int64_t)k0 * sak, sam, sak, B + (int64_t)k0 * sbk, sbk, sbn, M, N, K - k0, m0, n0, acc, lds);
    gemm_tile_store<0>(C, ldc, M, N, m0, n0, acc);
}

// M (n x n, ld = n): tiles above the diagonal are mirrored into the tiles below it; inside a diagonal tile the two
// independently computed halves are averaged (as symmetrize_kernel does for a full product)
__global__ void mirror_upper_kernel(double* __restrict__ M, int n) {
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (i <= j) continue;
        if ((i >> 6) == (j >> 6)) {
            const double a = M[(int64_t)i * n + j], b = M[(int64_t)j * n + i];
            const double s2 = 0.5 * (a + b);
            M[(int64_t)i * n + j] = s2;
            M[(int64_t)j * n + i] = s2;
        } else {
            M[(int64_t)i * n + j] = M[(int64_t)j * n + i];
        }
    }
}

int launch_gemm(const double* A, int64_t sam, int64_t sak, const double* B, int64_t sbk, int64_t sbn, double* C,
                int64_t ldc, int M, int N, int K, hipStream_t st) {
    if (M <= 0 || N <= 0) return TISE_OK;
    hipLaunchKernelGGL(gemm_f64_kernel, dim3(ceil_div(N, GT_BN), ceil_div(M, GT_BM)), dim3(256), 0, st, A, sam, sak, B,
                       sbk, sbn, C, ldc, M, N, K);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

__global__ void axpy_kernel(double* __restrict__ y, const double* __restrict__ x, double a, int64_t n) {
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += (int64_t)gridDim.x * blockDim.x)
        y[e] += a * x[e];
}

// M <- (M + M^T) / 2, n x n, ld = n
__global__ void symmetrize_kernel(double* __restrict__ M, int n) {
    const int64_t total = (int64_t)n * n;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / n), j = (int)(e % n);
        if (i < j) {
            const double a = M[(int64_t)i * n + j], b = M[(int64_t)j * n + i];
            const double s = 0.5 * (a + b);
            M[(int64_t)i * n + j] = s;
            M[(int64_t)j * n + i] = s;
        }
    }
}

// ------------------------------------------------------------------ Householder tridiagonalisation
// Single-workgroup step kernel for column k (see file header):
//   k > 0 : finish w_{k-1} = tau*p - (tau/2)(tau p.v) v from the matvec partials, apply reflector k-1
//           to column k on the fly (x_i = A[k][i] - v_i w_k - w_i v_k), td[k] = x_k
//   k = 0 : x = row 0
//   then generate reflector k from x[k+1 .. n-1]  (LAPACK dlarfg): te[k] = beta, tau, v (v[k+1] = 1).
__global__ __launch_bounds__(1024) void sytrd_step_kernel(const double* __restrict__ A, int n, int k,
                                                          const double* __restrict__ partial,
                                                          const double* __restrict__ vprev, double* __restrict__ w,
                                                          double* __restrict__ vnew, double* __restrict__ td,
                                                          double* __restrict__ te, FrState* __restrict__ st) {
    __shared__ double s_val[16];
    __shared__ double s_b[2];
    const int tid = threadIdx.x;
    if (k > 0) {
        const double tau = st->tau;                 // tau of reflector k-1
        const int rc0 = k / SY_RC;                  // first live row chunk of the matvec (rows >= k)
        const int nrc = (n + SY_RC - 1) / SY_RC;
        double dot = 0.0;
        for (int i = k + tid; i < n; i += blockDim.x) {
            double p = 0.0;
            for (int rc = rc0; rc < nrc; ++rc) p += partial[(int64_t)rc * n + i];
            p *= tau;
            w[i] = p;                               // provisional: tau * A v
            dot += p * vprev[i];
        }
        dot = block_sum(dot, s_val);
        const double alpha = -0.5 * tau * dot;
        for (int i = k + tid; i < n; i += blockDim.x) w[i] += alpha * vprev[i];
        __syncthreads();
        if (tid == 0) { s_b[0] = w[k]; s_b[1] = vprev[k]; }
        __syncthreads();
    }
    // x_i for i >= k (row k of the symmetric matrix, reflector k-1 applied on the fly)
    double xs = 0.0;       // sum of squares over i >= k+2
    for (int i = k + tid; i < n; i += blockDim.x) {
        double x = A[(int64_t)k * n + i];
        if (k > 0) x -= __dadd_rn(__dmul_rn(vprev[i], s_b[0]), __dmul_rn(w[i], s_b[1]));
        vnew[i] = x;                                // stash
        if (i >= k + 2) xs += x * x;
    }
    __syncthreads();
    xs = block_sum(xs, s_val);
    if (tid == 0) {
        td[k] = vnew[k];
        const double x0 = vnew[k + 1];
        double beta, tau, scale;
        if (xs == 0.0 || !isfinite(xs)) { tau = 0.0; beta = x0; scale = 0.0; }
        else {
            const double nrm = sqrt(x0 * x0 + xs);
            beta = -copysign(nrm, x0);
            tau = (beta - x0) / beta;
            scale = 1.0 / (x0 - beta);
        }
        te[k] = beta;
        st->tau_prev = st->tau;
        st->tau = tau;
        s_b[0] = scale;
    }
    __syncthreads();
    const double scale = s_b[0];
    for (int i = k + tid; i < n; i += blockDim.x) {
        if (i == k) vnew[i] = 0.0;
        else if (i == k + 1) vnew[i] = 1.0;
        else vnew[i] *= scale;
    }
}

// Fused trailing update + next matvec.  256 threads = 64 columns x 4 row phases; a workgroup covers a
// SY_RC x SY_CC tile with a FIXED tile -> blockIdx map, so a tile is revisited by the same XCD every step and
// stays in that XCD's L2 (the whole matrix is 33.5 MB against 32 MB of aggregate L2).
//   rows/cols >= k+1:  A[j][c] -= vprev[j] w[c] + w[j] vprev[c]      (reflector k-1; skipped for k == 0)
//   partial[rc][c] = sum_{j in chunk rc, j >= k+1} A[j][c] * vcur[j]  (matvec for reflector k)
// Lanes run along a row (coalesced 512-B segments); by symmetry the column sums this produces are the
// matrix-vector product, so no cross-lane reduction is needed.
__global__ __launch_bounds__(256) void sytrd_update_matvec_kernel(double* __restrict__ A, int n, int k,
                                                                  const double* __restrict__ vprev,
                                                                  const double* __restrict__ w,
                                                                  const double* __restrict__ vcur,
                                                                  double* __restrict__ partial) {
    const int rc = blockIdx.y;
    const int r0 = rc * SY_RC;
    const int cbase = blockIdx.x * SY_CC;
    if (min(r0 + SY_RC, n) <= k + 1 || cbase + SY_CC <= k + 1) return;   // dead tile
    __shared__ double s_vp[SY_RC], s_w[SY_RC], s_vc[SY_RC];
    __shared__ double s_part[4][SY_CC];
    {
        const int j = r0 + threadIdx.x;
        const bool ok = j < n && j >= k + 1;
        s_vp[threadIdx.x] = (ok && k > 0) ? vprev[j] : 0.0;
        s_w[threadIdx.x] = (ok && k > 0) ? w[j] : 0.0;
        s_vc[threadIdx.x] = ok ? vcur[j] : 0.0;
    }
    __syncthreads();
    const int lc = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = cbase + lc;
    const bool live = c < n && c >= k + 1;
    const int j0 = max(r0 + ph * 64, k + 1);
    const int j1 = min(r0 + ph * 64 + 64, n);
    double acc = 0.0;
    if (live) {
        if (k > 0) {
            const double vpc = vprev[c], wc = w[c];
            for (int j = j0; j < j1; ++j) {
                const int jj = j - r0;
                double a = A[(int64_t)j * n + c];
                a -= __dadd_rn(__dmul_rn(s_vp[jj], wc), __dmul_rn(s_w[jj], vpc));
                A[(int64_t)j * n + c] = a;
                acc += a * s_vc[jj];
            }
        } else {
            for (int j = j0; j < j1; ++j) acc += A[(int64_t)j * n + c] * s_vc[j - r0];
        }
    }
    s_part[ph][lc] = acc;
    __syncthreads();
    if (ph == 0 && live) partial[(int64_t)rc * n + c] = ((s_part[0][lc] + s_part[1][lc]) + s_part[2][lc]) + s_part[3][lc];
}

// One launch per column (default for n <= SYF_MAX_N).  The two-launch scheme above spends 7 us per column in a
// single-workgroup kernel and 15 us in an update kernel whose 256 workgroups each walk 64 rows serially.  Here
// EVERY workgroup first repeats the O(n) bookkeeping of column k for itself -- finish w_{k-1} from the raw
// product p = A v_{k-1} and the partial dot products the previous launch left, form row k with reflector k-1
// applied on the fly, generate reflector k (identical arithmetic in every workgroup, so all agree bit for bit) --
// keeping w_{k-1}, v_{k-1}, v_k in LDS, and then owns whole ROWS of the trailing block: wave = one row at a
// time, lanes along the row (coalesced 512-byte segments),
//     A[j][c] -= v_{k-1}[j] w[c] + w[j] v_{k-1}[c]          (c, j >= k+1; skipped for k == 0)
//     p_k[j]   = sum_c A[j][c] v_k[c]                        (wave reduction; complete per row: no partials)
// plus the workgroup's share of p_k . v_k.  Rows are dealt round-robin over the LIVE rows, so late columns keep
// every launched workgroup busy.  Exact symmetry is kept the same way as above (__dmul_rn/__dadd_rn).
__global__ __launch_bounds__(256) void sytrd_fused_kernel(double* __restrict__ A, int n, int k,
                                                          const double* __restrict__ vprev_g, double* __restrict__ vcur_g,
                                                          const double* __restrict__ p_prev, double* __restrict__ p_cur,
                                                          const double* __restrict__ sd_prev, int g_prev,
                                                          double* __restrict__ sd_cur, double* __restrict__ td,
                                                          double* __restrict__ te, double* __restrict__ tau_arr) {
    extern __shared__ double smem[];
    double* s_w = smem;
    double* s_vp = smem + n;
    double* s_vk = smem + 2 * (size_t)n;
    __shared__ double s_val[16];
    const int tid = threadIdx.x;
    double wk = 0.0, vpk = 0.0;
    if (k > 0) {
        const double tau = tau_arr[k - 1];
        double d = 0.0;
        for (int g = tid; g < g_prev; g += 256) d += sd_prev[g];
        d = block_sum(d, s_val);                    // p . v_{k-1}
        const double alpha = -0.5 * tau * (tau * d);
        for (int i = k + tid; i < n; i += 256) {
            const double vp = vprev_g[i];
            s_vp[i] = vp;
            s_w[i] = tau * p_prev[i] + alpha * vp;
        }
        __syncthreads();
        wk = s_w[k];
        vpk = s_vp[k];
    }
    double xs = 0.0;                                // sum of squares over i >= k+2
    for (int i = k + tid; i < n; i += 256) {
        double x = A[(int64_t)k * n + i];
        if (k > 0) x -= __dadd_rn(__dmul_rn(s_vp[i], wk), __dmul_rn(s_w[i], vpk));
        s_vk[i] = x;
        if (i >= k + 2) xs += x * x;
    }
    xs = block_sum(xs, s_val);                      // (contains the barrier that publishes s_vk)
    const double xk = s_vk[k], x0 = s_vk[k + 1];
    double beta, tau_k, scale;
    if (xs == 0.0 || !isfinite(xs)) { tau_k = 0.0; beta = x0; scale = 0.0; }
    else {
        const double nrm = sqrt(x0 * x0 + xs);
        beta = -copysign(nrm, x0);
        tau_k = (beta - x0) / beta;
        scale = 1.0 / (x0 - beta);
    }
    __syncthreads();
    for (int i = k + tid; i < n; i += 256) {
        const double v = (i == k) ? 0.0 : (i == k + 1) ? 1.0 : s_vk[i] * scale;
        s_vk[i] = v;
        if (blockIdx.x == 0) vcur_g[i] = v;
    }
    if (blockIdx.x == 0 && tid == 0) { td[k] = xk; te[k] = beta; tau_arr[k] = tau_k; }
    __syncthreads();
    // ---- own rows of the trailing block ---------------------------------------------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int nlive = n - (k + 1);
    const int c0 = (k + 1) & ~15;                   // 128-byte aligned start of the row segment
    double pd = 0.0;
    for (int r = blockIdx.x * 4 + wave; r < nlive; r += gridDim.x * 4) {
        const int j = k + 1 + r;
        double* row = A + (int64_t)j * n;
        const double vpj = k > 0 ? s_vp[j] : 0.0, wj = k > 0 ? s_w[j] : 0.0;
        double acc = 0.0;
        for (int cb = c0; cb < n; cb += 256) {      // four 64-lane segments per trip: loads first, then the arithmetic
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + 64 * u + lane;
                a[u] = (c > k && c < n) ? row[c] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int c = cb + 64 * u + lane;
                if (c > k && c < n) {
                    double v = a[u];
                    if (k > 0) {
                        v -= __dadd_rn(__dmul_rn(vpj, s_w[c]), __dmul_rn(wj, s_vp[c]));
                        row[c] = v;
                    }
                    acc += v * s_vk[c];
                }
            }
        }
        acc = wave_sum(acc);
        if (lane == 0) {
            p_cur[j] = acc;
            pd += acc * s_vk[j];
        }
    }
    if (lane == 0) s_val[wave] = pd;
    __syncthreads();
    if (tid == 0) sd_cur[blockIdx.x] = ((s_val[0] + s_val[1]) + s_val[2]) + s_val[3];
}

// n <= 2048 form of the kernel above (the metric's d): the launch is latency-, not bandwidth-bound for most
// columns (measured 13.4 us per column with three dependent load phases and five barriers in the bookkeeping),
// so here every global value the bookkeeping needs -- tau, the partial dots, v_{k-1}, p, row k, and the first
// segments of the wave's own trailing row -- is requested up front (8 elements per thread per vector), every
// wave sums the partial dots for itself, and two barriers remain.  Rows map to workgroups by a FIXED rule
// (row j -> workgroup j / 4, one row per wave), so a row is revisited by the same XCD in every launch and the
// 33.5 MB matrix stays spread over the eight 4 MB L2s; the grid starts at the first workgroup (rounded down to
// a multiple of 8, which keeps the workgroup -> XCD map) that still owns a live row.
// WAVES x RPW (round 5): rows per workgroup.  The per-column trace of the 4-wave form (profiles/r02f_sytrd_per_column_trace.txt,
// r05h_sytrd_per_column_rows{4,8}.txt) falls almost LINEARLY from 14.6 us at k = 0 to 5.9 us; a fit a + b m + c m^2 over the
// live rows m = n - k gives a = 5.6 us (x 2047 launches = 11.4 ms: the dependent-launch chain), b = 2.5 ns per row (5.2 ms) and
// c m^2 = 3.9 us at k = 0 (2.7 ms in all: the only part that is matrix traffic -- rounds 2-4 read all 7.8 ms above the floor as
// bandwidth).  Going from 4 to 8 rows per workgroup takes b to 1.6 ns: b = 0.7 ns per row + 7.2 ns per WORKGROUP (dispatch and
// the O(n) bookkeeping every workgroup repeats).  The row loop reads LDS eight elements at a time with selects instead of a
// branch and two exposed LDS round trips per element; chunks behind the row's end are skipped by a uniform branch.
template <int WAVES, int RPW>
__global__ __launch_bounds__(64 * WAVES) void sytrd_fused8_kernel(double* __restrict__ A, int n, int k, int wg0,
                                                           const double* __restrict__ vprev_g, double* __restrict__ vcur_g,
                                                           const double* __restrict__ p_prev, double* __restrict__ p_cur,
                                                           const double* __restrict__ sd_prev, int sd_lo, int sd_hi,
                                                           double* __restrict__ sd_cur, double* __restrict__ td,
                                                           double* __restrict__ te, double* __restrict__ tau_arr) {
    constexpr int NT = 64 * WAVES, NU = 2048 / NT;                 // threads, bookkeeping elements per thread
    __shared__ double s_w[2048], s_vp[2048], s_vk[2048];
    __shared__ double s_val[WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wg = wg0 + blockIdx.x;
    const int jb = (wg * WAVES + wave) * RPW;                     // this wave's RPW consecutive rows
    const int c0 = (k + 1) & ~15;
    // ---- every global request first --------------------------------------------------------------
    double xr[NU], vpr[NU], ppr[NU], ar[RPW][32];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = k + tid + NT * u;
        xr[u] = i < n ? A[(int64_t)k * n + i] : 0.0;
        vpr[u] = (k > 0 && i < n) ? vprev_g[i] : 0.0;
        ppr[u] = (k > 0 && i < n) ? p_prev[i] : 0.0;
    }
    // the wave's whole trailing row (<= 32 segments of 64 doubles) is requested now: its latency hides under the
    // bookkeeping, and the row is never the limiter of memory-level parallelism (4 requests per lane in flight per
    // trip measured 1.7 us per 256-column trip: 19 us at k = 0)
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int j = jb + r;
        const bool row_live = j > k && j < n;
        const double* row = A + (int64_t)(row_live ? j : k) * n;
#pragma unroll
        for (int u = 0; u < 32; ++u) {
            const int c = c0 + 64 * u + lane;
            ar[r][u] = (row_live && c > k && c < n) ? row[c] : 0.0;
        }
    }
    double tau = 0.0, alpha = 0.0, wk = 0.0, vpk = 0.0;
    if (k > 0) {
        tau = tau_arr[k - 1];
        double d = 0.0;
        for (int g = sd_lo + lane; g < sd_hi; g += 64) d += sd_prev[g];
        d = wave_sum(d);                                          // p . v_{k-1}: every wave for itself, same order
        alpha = -0.5 * tau * (tau * d);
        vpk = vprev_g[k];
        wk = tau * p_prev[k] + alpha * vpk;
    }
    // ---- w_{k-1}, v_{k-1} -> LDS; row k with reflector k-1 applied; its tail norm -----------------
    double xs = 0.0;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = k + tid + NT * u;
        if (i < n) {
            double x = xr[u];
            if (k > 0) {
                const double w = tau * ppr[u] + alpha * vpr[u];
                s_w[i] = w;
                s_vp[i] = vpr[u];
                x -= __dadd_rn(__dmul_rn(vpr[u], wk), __dmul_rn(w, vpk));
            }
            xr[u] = x;
            if (i >= k + 2) xs += x * x;
        }
    }
    xs = wave_sum(xs);
    if (lane == 0) s_val[wave] = xs;
    // x_k and x_{k+1} sit in thread 0 (u = 0) and thread 1 (u = 0): publish them with the same barrier
    __shared__ double s_x[2];
    if (tid < 2 && k + tid < n) s_x[tid] = xr[0];
    __syncthreads();
    xs = s_val[0];
#pragma unroll
    for (int q = 1; q < WAVES; ++q) xs += s_val[q];              // fixed order: every workgroup gets the same bits
    const double xk = s_x[0], x0 = s_x[1];
    double beta, tau_k, scale;
    if (xs == 0.0 || !isfinite(xs)) { tau_k = 0.0; beta = x0; scale = 0.0; }
    else {
        const double nrm = sqrt(x0 * x0 + xs);
        beta = -copysign(nrm, x0);
        tau_k = (beta - x0) / beta;
        scale = 1.0 / (x0 - beta);
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int i = k + tid + NT * u;
        if (i < n) {
            const double v = (i == k) ? 0.0 : (i == k + 1) ? 1.0 : xr[u] * scale;
            s_vk[i] = v;
            if (blockIdx.x == 0) vcur_g[i] = v;
        }
    }
    if (blockIdx.x == 0 && tid == 0) { td[k] = xk; te[k] = beta; tau_arr[k] = tau_k; }
    __syncthreads();
    // ---- the wave's row of the trailing block -------------------------------------------------------
    double pd = 0.0;
#pragma unroll
    for (int r = 0; r < RPW; ++r) {
        const int j = jb + r;
        if (j > k && j < n) {                                       // wave-uniform
            double* row = A + (int64_t)j * n;
            const double vpj = k > 0 ? s_vp[j] : 0.0, wj = k > 0 ? s_w[j] : 0.0;
            double acc = 0.0;
            // Branch-free, LDS reads batched eight elements at a time (round 5).  The first form -- `if (c > k && c < n)`
            // around every element -- compiled to an exec-mask branch and two exposed LDS round trips PER ELEMENT (ds_read2 ->
            // s_waitcnt lgkmcnt(0) -> ds_read -> s_waitcnt): 32 elements x ~200 cycles = the 3.8 ns per live row that the
            // per-column trace shows as its linear term.  Reads use a clamped index (always inside the arrays), elements outside
            // (k, n) are removed by selects, only the store keeps its predicate.
            if (k > 0) {                                            // uniform: two straight-line bodies, no per-element control flow
#pragma unroll
                for (int ub = 0; ub < 32; ub += 8) {
                    if (c0 + 64 * ub >= n) break;                   // uniform: the chunk lies behind the row's end
                    double sw[8], sp[8], sk[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int cc = min(c0 + 64 * (ub + q) + lane, 2047);
                        sk[q] = s_vk[cc]; sw[q] = s_w[cc]; sp[q] = s_vp[cc];
                    }
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = c0 + 64 * (ub + q) + lane;
                        const bool ok = c > k && c < n;
                        const double v = ar[r][ub + q] - __dadd_rn(__dmul_rn(vpj, sw[q]), __dmul_rn(wj, sp[q]));
                        if (ok) row[c] = v;
                        const double t = v * sk[q];
                        acc += ok ? t : 0.0;
                    }
                }
            } else {
#pragma unroll
                for (int ub = 0; ub < 32; ub += 8) {
                    if (c0 + 64 * ub >= n) break;
                    double sk[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) sk[q] = s_vk[min(c0 + 64 * (ub + q) + lane, 2047)];
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        const int c = c0 + 64 * (ub + q) + lane;
                        const double t = ar[r][ub + q] * sk[q];
                        acc += (c > k && c < n) ? t : 0.0;
                    }
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) {
                p_cur[j] = acc;
                pd += acc * s_vk[j];
            }
        }
    }
    __syncthreads();                                                // s_val reuse
    if (lane == 0) s_val[wave] = pd;
    __syncthreads();
    if (tid == 0) {
        double t = s_val[0];
#pragma unroll
        for (int q = 1; q < WAVES; ++q) t += s_val[q];
        sd_cur[wg] = t;
    }
}

__global__ void sytrd_last_kernel(const double* __restrict__ A, int n, double* __restrict__ td) {
    if (threadIdx.x == 0 && blockIdx.x == 0) td[n - 1] = A[(int64_t)(n - 1) * n + (n - 1)];
}

// ------------------------------------------------------------------ tridiagonal eigenvalues by bisection
__global__ __launch_bounds__(1024) void gershgorin_kernel(const double* __restrict__ td, const double* __restrict__ te,
                                                          int n, FrState* __restrict__ st) {
    __shared__ double s_val[16];
    double lo = INFINITY, hi = -INFINITY, e2max = 0.0;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const double el = i > 0 ? fabs(te[i - 1]) : 0.0;
        const double er = i < n - 1 ? fabs(te[i]) : 0.0;
        lo = fmin(lo, td[i] - el - er);
        hi = fmax(hi, td[i] + el + er);
        e2max = fmax(e2max, er * er);
    }
    // block max / min through wave shuffles
    lo = -wave_max(-lo); hi = wave_max(hi); e2max = wave_max(e2max);
    __shared__ double s_lo[16], s_hi[16];
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_lo[wave] = lo; s_hi[wave] = hi; s_val[wave] = e2max; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < (int)(blockDim.x >> 6); ++i) {
            lo = fmin(lo, s_lo[i]); hi = fmax(hi, s_hi[i]); e2max = fmax(e2max, s_val[i]);
        }
        const double nrm = fmax(fabs(lo), fabs(hi));
        const double pad = 2.0 * nrm * 2.220446049250313e-16 * n + 2.0 * 2.2250738585072014e-308;
        st->glo = lo - pad;
        st->ghi = hi + pad;
        st->pivmin = 2.2250738585072014e-308 * fmax(1.0, e2max);
    }
}

// number of eigenvalues of T that are < x  (LAPACK dlaebz-style Sturm count), for NP shifts at once: the recurrence
//     q_i = d_i - x - e_{i-1}^2 / q_{i-1}
// is one long dependent chain per shift (n steps); a lane can carry NP independent chains (default 1: see
// launch_bisect) and the division is a v_rcp_f64 plus one Newton step (3 dependent instructions instead of the
// ~10 of the correctly rounded division: 2.65 -> 1.83 ms at n = 2048, same eigenvalues to the last digit printed;
// the count is insensitive to the last bit of q except when q is within rounding of zero, where pivmin decides).
template <int NP>
__device__ __forceinline__ void sturm_count_multi(const double* __restrict__ td, const double* __restrict__ te, int n,
                                                  const double (&x)[NP], double pivmin, int (&cnt)[NP]) {
    double q[NP];
    const double d0 = td[0];
#pragma unroll
    for (int u = 0; u < NP; ++u) {
        q[u] = d0 - x[u];
        cnt[u] = 0;
        if (q[u] <= pivmin) { ++cnt[u]; q[u] = fmin(q[u], -pivmin); }
    }
    for (int i = 1; i < n; ++i) {
        const double e = te[i - 1];
        const double e2 = e * e, di = td[i];
#pragma unroll
        for (int u = 0; u < NP; ++u) {
            double r = __builtin_amdgcn_rcp(q[u]);
            r = fma(fma(-q[u], r, 1.0), r, r);
            q[u] = (di - x[u]) - e2 * r;
            if (q[u] <= pivmin) { ++cnt[u]; q[u] = fmin(q[u], -pivmin); }
        }
    }
}

// One wave per eigenvalue, (64 NP)-way multisection: every lane evaluates the Sturm count at NP shifts of its own
// inside the current bracket (probe index = u * 64 + lane, i.e. 64 NP equally spaced shifts), ballots pick
// the sub-interval that contains the m-th eigenvalue.  8 bits per pass instead of 1 => ~7 passes over the
// tridiagonal recurrence instead of ~55, and n waves instead of n threads keep every CU busy.  td/te are read at
// wave-uniform addresses (scalar loads).
template <int BIS_NP>
__global__ __launch_bounds__(256) void bisect_kernel(const double* __restrict__ td, const double* __restrict__ te, int n,
                                                     const FrState* __restrict__ st, double* __restrict__ eig) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= n) return;
    double lo = st->glo, hi = st->ghi;
    const double pivmin = st->pivmin;
    // absolute accuracy eps * ||T|| is all the trace of the square root needs (and all that the
    // tridiagonalisation preserved)
    const double tol = 2.220446049250313e-16 * fmax(fabs(lo), fabs(hi)) + 2.0 * pivmin;
    constexpr int NPROBE = 64 * BIS_NP;
    for (int it = 0; it < 40; ++it) {
        const double width = hi - lo;
        if (!(width > tol)) break;
        double x[BIS_NP];
        int cnt[BIS_NP];
#pragma unroll
        for (int u = 0; u < BIS_NP; ++u) x[u] = lo + width * ((double)(u * 64 + lane + 1) * (1.0 / (NPROBE + 1)));
        sturm_count_multi<BIS_NP>(td, te, n, x, pivmin, cnt);
        // first probe (in index order u * 64 + lane) whose count reaches m + 1; counts are monotone in the probe index
        int first = NPROBE;
#pragma unroll
        for (int u = BIS_NP - 1; u >= 0; --u) {
            const unsigned long long mask = __ballot(cnt[u] >= m + 1);
            if (mask != 0ull) first = u * 64 + (__ffsll((long long)mask) - 1);
        }
        // x of probe j: every lane can compute it (same formula)
        const double nhi = first < NPROBE ? lo + width * ((double)(first + 1) * (1.0 / (NPROBE + 1))) : hi;
        const double nlo = first > 0 ? lo + width * ((double)first * (1.0 / (NPROBE + 1))) : lo;
        if (!(nhi - nlo < width)) break;         // no progress at the resolution of the doubles
        lo = nlo;
        hi = nhi;
    }
    if (lane == 0) eig[m] = 0.5 * (lo + hi);
}

// ------------------------------------------------------------------ final combination
__global__ __launch_bounds__(1024) void frechet_finish_kernel(const double* __restrict__ mu1, const double* __restrict__ s1,
                                                              const double* __restrict__ mu2, const double* __restrict__ s2,
                                                              int d, double off, const double* __restrict__ eig, int r,
                                                              const FrState* __restrict__ st, double* __restrict__ out) {
    __shared__ double s_val[16];
    double dd = 0.0, t1 = 0.0, t2 = 0.0, ts = 0.0, neg = 0.0;
    for (int i = threadIdx.x; i < d; i += blockDim.x) {
        const double df = mu1[i] - mu2[i];
        dd += df * df;
        t1 += s1[(int64_t)i * d + i];
        t2 += s2[(int64_t)i * d + i];
    }
    for (int i = threadIdx.x; i < r; i += blockDim.x) {
        const double l = eig[i];
        if (l > 0.0) ts += sqrt(l);
        else if (l < 0.0) neg += 1.0;
    }
    dd = block_sum(dd, s_val);
    t1 = block_sum(t1, s_val);
    t2 = block_sum(t2, s_val);
    ts = block_sum(ts, s_val);
    neg = block_sum(neg, s_val);
    if (threadIdx.x == 0) {
        (void)off;   // the reference's retry offsets only the matrices inside sqrtm, not the traces (:160 vs :171)
        int flags = 0;
        if (st->nonfinite || !isfinite(ts) || !isfinite(dd) || !isfinite(t1) || !isfinite(t2)) flags |= TISE_FLAG_NONFINITE;
        if (r < d) flags |= TISE_FLAG_RANK_DEFICIENT;
        out[0] = dd + t1 + t2 - 2.0 * ts;
        out[1] = ts;
        out[2] = dd;
        out[3] = t1;
        out[4] = t2;
        out[5] = (double)r;
        out[6] = neg;
        out[7] = (double)flags;
    }
}

// ------------------------------------------------------------------ host drivers
int run_pchol(tise_frechet* h, const double* S, double off, int* rank_out, hipStream_t st);

// factor of S (+ off I) into h->lt: the unpivoted fast path when it succeeds (full numerical rank), else the pivoted one
int run_chol(tise_frechet* h, const double* S, double off, int* rank_out, hipStream_t st) {
    const int d = h->d;
    const bool pivoted_only = getenv("TISE_CHOL_PIVOTED") != nullptr;             // A/B switch, read per call (tests, tools/frechet_probe.py)
    if (d <= 2048 && d >= CHB && !pivoted_only) {
        int* fail = h->chosen;                                     // one int of the pivoted path's scratch, rewritten by it anyway
        double* maxdiag = h->diag;
        hipLaunchKernelGGL(pchol_copy_kernel, dim3(2048), dim3(256), 0, st, S, d, off, h->t1);
        hipLaunchKernelGGL(chol_maxdiag_kernel, dim3(1), dim3(1024), 0, st, h->t1, d, maxdiag, fail);
        TISE_LAUNCH_CHECK();
        double md = 0.0;
        int f0 = 0;
        TISE_HIP_CHECK(hipMemcpyAsync(&md, maxdiag, sizeof(double), hipMemcpyDeviceToHost, st));
        TISE_HIP_CHECK(hipMemcpyAsync(&f0, fail, sizeof(int), hipMemcpyDeviceToHost, st));
        TISE_HIP_CHECK(hipStreamSynchronize(st));
        if (!f0 && md > 0.0) {
            const double tol = 1e-12 * md;
            hipLaunchKernelGGL(chol_zero_upper_kernel, dim3(2048), dim3(256), 0, st, h->lt, d);
            for (int k0 = 0; k0 < d; k0 += CHB) {
                const int npanel = ceil_div(d - k0, CHB);
                hipLaunchKernelGGL(chol_panel_kernel, dim3(npanel), dim3(256), 0, st, h->t1, d, k0, tol, h->lt, fail);
                if (npanel > 1)
                    hipLaunchKernelGGL(chol_trail_kernel, dim3(npanel - 1, npanel - 1), dim3(256), 0, st, h->lt, d, k0, fail, h->t1);
            }
            hipLaunchKernelGGL(chol_finish_kernel, dim3(1), dim3(64), 0, st, fail, d, h->st);
            TISE_LAUNCH_CHECK();
            TISE_HIP_CHECK(hipMemcpyAsync(&f0, fail, sizeof(int), hipMemcpyDeviceToHost, st));
            TISE_HIP_CHECK(hipStreamSynchronize(st));
            if (!f0) { *rank_out = d; h->lt_tri = 1; return TISE_OK; }
        }
    }
    return run_pchol(h, S, off, rank_out, st);
}

int run_pchol(tise_frechet* h, const double* S, double off, int* rank_out, hipStream_t st) {
    const int d = h->d;
    h->lt_tri = 0;                                             // pivoted: the columns of L come in pivot order
    if (d <= 2048) {                                   // blocked path: working copy lives in h->t1
        FrState hs;
        hipLaunchKernelGGL(pchol_copy_kernel, dim3(2048), dim3(256), 0, st, S, d, off, h->t1);
        hipLaunchKernelGGL(pchol_init_kernel, dim3(1), dim3(1024), 0, st, h->t1, d, 0.0, h->diag, h->chosen, h->st);
        TISE_LAUNCH_CHECK();
        const dim3 tgrid(ceil_div(d, GT_BN), ceil_div(d, GT_BM));
        int nblk = 0;
        for (int k0 = 0; k0 < d; k0 += PCB_NB, ++nblk) {
            if (d <= 1024)
                hipLaunchKernelGGL(pchol_panel_kernel<1>, dim3(1), dim3(1024), 0, st, h->t1, d, k0, h->lt, h->diag, h->chosen, h->st);
            else
                hipLaunchKernelGGL(pchol_panel_kernel<2>, dim3(1), dim3(1024), 0, st, h->t1, d, k0, h->lt, h->diag, h->chosen, h->st);
            if (k0 + PCB_NB < d)
                hipLaunchKernelGGL(pchol_trail_kernel, tgrid, dim3(256), 0, st, h->lt, d, k0, h->st, h->t1);
            if ((nblk & 7) == 7 && k0 + PCB_NB < d) {   // cheap early-out for rank-deficient inputs
                TISE_HIP_CHECK(hipMemcpyAsync(&hs, h->st, sizeof(FrState), hipMemcpyDeviceToHost, st));
                TISE_HIP_CHECK(hipStreamSynchronize(st));
                if (hs.done) break;
            }
        }
        TISE_LAUNCH_CHECK();
        TISE_HIP_CHECK(hipMemcpyAsync(&hs, h->st, sizeof(FrState), hipMemcpyDeviceToHost, st));
        TISE_HIP_CHECK(hipStreamSynchronize(st));
        *rank_out = hs.rank;
        return TISE_OK;
    }
    hipLaunchKernelGGL(pchol_init_kernel, dim3(1), dim3(1024), 0, st, S, d, off, h->diag, h->chosen, h->st);
    TISE_LAUNCH_CHECK();
    const dim3 pgrid(ceil_div(d, 64), PCHOL_NP);
    FrState host_state;
    for (int k = 0; k < d; ++k) {
        hipLaunchKernelGGL(pchol_partial_kernel, pgrid, dim3(256), 0, st, h->lt, d, k, h->st, h->partial);
        hipLaunchKernelGGL(pchol_finish_kernel, dim3(1), dim3(1024), 0, st, S, d, k, off, h->partial, h->lt, h->diag,
                           h->colbuf, h->chosen, h->st);
        if ((k & 127) == 127 && k + 1 < d) {     // cheap early-out for rank-deficient inputs
            TISE_HIP_CHECK(hipMemcpyAsync(&host_state, h->st, sizeof(FrState), hipMemcpyDeviceToHost, st));
            TISE_HIP_CHECK(hipStreamSynchronize(st));
            if (host_state.done) break;
        }
    }
    TISE_LAUNCH_CHECK();
    TISE_HIP_CHECK(hipMemcpyAsync(&host_state, h->st, sizeof(FrState), hipMemcpyDeviceToHost, st));
    TISE_HIP_CHECK(hipStreamSynchronize(st));
    *rank_out = host_state.rank;
    return TISE_OK;
}

void launch_bisect(tise_frechet* h, int n, hipStream_t st) {
    // chains per lane: measured 1.83 / 2.61 / 4.15 ms for 1 / 2 / 4 at n = 2048 (profiles/r02i_bisect_np.txt): with two
    // waves per SIMD the recurrence is bound by fp64 issue, not by its latency, so extra chains only add work
    static const int np = getenv("TISE_BISECT_NP") ? atoi(getenv("TISE_BISECT_NP")) : 1;
    const dim3 grid(ceil_div(n, 4)), block(256);
    if (np == 1) hipLaunchKernelGGL(bisect_kernel<1>, grid, block, 0, st, h->td, h->te, n, h->st, h->eig);
    else if (np == 4) hipLaunchKernelGGL(bisect_kernel<4>, grid, block, 0, st, h->td, h->te, n, h->st, h->eig);
    else hipLaunchKernelGGL(bisect_kernel<2>, grid, block, 0, st, h->td, h->te, n, h->st, h->eig);
}

// eigenvalues of the symmetric n x n matrix in h->m (ld = n, destroyed) -> h->eig[0..n)
int run_eigvalsh_inplace(tise_frechet* h, int n, hipStream_t st) {
    if (n <= 0) return TISE_OK;
    double* A = h->m;
    if (n == 1) {
        TISE_HIP_CHECK(hipMemcpyAsync(h->eig, A, sizeof(double), hipMemcpyDeviceToDevice, st));
        return TISE_OK;
    }
    static const bool two_launch = getenv("TISE_SYTRD_TWO_LAUNCH") != nullptr;     // A/B switch (tests, tools/frechet_probe.py)
    if (n <= 2048 && !two_launch && getenv("TISE_SYTRD_FUSED_GENERIC") == nullptr) {
        double* pb[2] = {h->fz, h->fz + h->d};
        double* sd[2] = {h->fz + 2 * (size_t)h->d, h->fz + 2 * (size_t)h->d + SYF_G};
        double* tau_arr = h->fz + 2 * (size_t)h->d + 2 * SYF_G;
        double* vv[2] = {h->va, h->vb};
        // rows per workgroup (sytrd_fused8_kernel<WAVES>): 16 by default, TISE_SYTRD_WAVES=4 / 8 for the A/B (4 = rounds 2-4)
        // TISE_SYTRD_ROWS = rows per workgroup: 8 (default, round 5: 8 waves x 1 row), 4 (rounds 2-4: 4 waves x 1 row), 16 (8 waves x 2
        // rows), 82 (4 waves x 2 rows = 8).  Measured at n = 2048 (profiles/r05g_frechet_rows_per_workgroup.txt): 4 -> 19.6 ms,
        // 8 -> 18.0, 82 -> 23.9, 16 -> 23.0 (a second row per wave doubles the wave's serial chain); 16 waves x 1 row needs 127
        // VGPRs and spills 78 (39.8 ms)
        static const int cfg = [] { const char* e = getenv("TISE_SYTRD_ROWS"); const int w = e ? atoi(e) : 8; return (w == 4 || w == 16 || w == 82) ? w : 8; }();
        const int waves = cfg == 82 ? 8 : cfg;                       // rows per workgroup
        const int wg_end = ceil_div(n, waves);                       // workgroup g owns rows waves * g .. waves * g + waves - 1
        const int xcd_mask = ~7;                                     // the grid starts on a multiple of 8 workgroups: a row stays on one XCD (its L2) for the whole solve
        int sd_lo = 0, sd_hi = 0;
        for (int k = 0; k <= n - 2; ++k) {
            const int wg0 = ((k + 1) / waves) & xcd_mask;            // first workgroup with a live row
#define TISE_SYTRD_LAUNCH(W, R)                                                                                     \
            hipLaunchKernelGGL((sytrd_fused8_kernel<W, R>), dim3(wg_end - wg0), dim3(64 * W), 0, st, A, n, k, wg0, vv[(k + 1) & 1], \
                               vv[k & 1], pb[(k + 1) & 1], pb[k & 1], sd[(k + 1) & 1], sd_lo, sd_hi, sd[k & 1], h->td,  \
                               h->te, tau_arr)
            if (cfg == 16) TISE_SYTRD_LAUNCH(8, 2);
            else if (cfg == 8) TISE_SYTRD_LAUNCH(8, 1);
            else if (cfg == 82) TISE_SYTRD_LAUNCH(4, 2);
            else TISE_SYTRD_LAUNCH(4, 1);
#undef TISE_SYTRD_LAUNCH
            sd_lo = wg0;
            sd_hi = wg_end;
        }
        TISE_LAUNCH_CHECK();
        hipLaunchKernelGGL(sytrd_last_kernel, dim3(1), dim3(64), 0, st, A, n, h->td);
        TISE_LAUNCH_CHECK();
        if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[3], st));
        hipLaunchKernelGGL(gershgorin_kernel, dim3(1), dim3(1024), 0, st, h->td, h->te, n, h->st);
        launch_bisect(h, n, st);
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    if (n <= SYF_MAX_N && !two_launch) {
        double* pb[2] = {h->fz, h->fz + h->d};
        double* sd[2] = {h->fz + 2 * (size_t)h->d, h->fz + 2 * (size_t)h->d + SYF_G};
        double* tau_arr = h->fz + 2 * (size_t)h->d + 2 * SYF_G;
        double* vv[2] = {h->va, h->vb};
        const size_t lds = 3 * (size_t)n * sizeof(double);
        if (lds > 48 * 1024)
            TISE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(sytrd_fused_kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        int g_prev = 0;
        for (int k = 0; k <= n - 2; ++k) {
            const int nlive = n - (k + 1);
            int g = ceil_div(nlive, 4);
            if (g > SYF_G) g = SYF_G;
            hipLaunchKernelGGL(sytrd_fused_kernel, dim3(g), dim3(256), lds, st, A, n, k, vv[(k + 1) & 1], vv[k & 1],
                               pb[(k + 1) & 1], pb[k & 1], sd[(k + 1) & 1], g_prev, sd[k & 1], h->td, h->te, tau_arr);
            g_prev = g;
        }
        TISE_LAUNCH_CHECK();
        // launch k = n-2 (trivial reflector, tau = 0) has applied reflector n-3 to A[n-1][n-1]
        hipLaunchKernelGGL(sytrd_last_kernel, dim3(1), dim3(64), 0, st, A, n, h->td);
        TISE_LAUNCH_CHECK();
        if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[3], st));
        hipLaunchKernelGGL(gershgorin_kernel, dim3(1), dim3(1024), 0, st, h->td, h->te, n, h->st);
        launch_bisect(h, n, st);
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    const dim3 ugrid(ceil_div(n, SY_CC), ceil_div(n, SY_RC));   // 32 x 8 = 256 workgroups at n = 2048
    double* vprev = h->va;
    double* vcur = h->vb;
    for (int k = 0; k <= n - 2; ++k) {
        hipLaunchKernelGGL(sytrd_step_kernel, dim3(1), dim3(1024), 0, st, A, n, k, h->partial, vprev, h->w, vcur, h->td,
                           h->te, h->st);
        hipLaunchKernelGGL(sytrd_update_matvec_kernel, ugrid, dim3(256), 0, st, A, n, k, vprev, h->w, vcur, h->partial);
        double* t = vprev; vprev = vcur; vcur = t;
    }
    TISE_LAUNCH_CHECK();
    hipLaunchKernelGGL(sytrd_last_kernel, dim3(1), dim3(64), 0, st, A, n, h->td);
    if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[3], st));
    hipLaunchKernelGGL(gershgorin_kernel, dim3(1), dim3(1024), 0, st, h->td, h->te, n, h->st);
    launch_bisect(h, n, st);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // namespace

extern "C" {

int tise_frechet_create(int d, tise_frechet_t** out) {
    if (d <= 0 || d > 8192 || !out) return TISE_ERR_INVALID_ARG;
    tise_frechet* h = new (std::nothrow) tise_frechet;
    if (!h) return TISE_ERR_INVALID_ARG;
    memset(h, 0, sizeof(*h));
    h->d = d;
    const size_t dd = (size_t)d * d;
    const size_t nchunk = (size_t)((d + SY_RC - 1) / SY_RC + 1) > (size_t)PCHOL_NP ? (size_t)((d + SY_RC - 1) / SY_RC + 1) : (size_t)PCHOL_NP;
    // one allocation, carved
    const size_t fz = 3 * (size_t)d + 2 * SYF_G;
    const size_t total = 3 * dd + nchunk * d + 8 * (size_t)d + fz + 64;
    double* base = nullptr;
    hipError_t e = hipMalloc((void**)&base, total * sizeof(double) + (size_t)d * sizeof(int) + sizeof(FrState) + 256);
    if (e != hipSuccess) { tise_set_last_hip_error((int)e); delete h; return TISE_ERR_HIP; }
    e = hipMemset(base, 0, total * sizeof(double) + (size_t)d * sizeof(int) + sizeof(FrState) + 256);
    if (e != hipSuccess) { tise_set_last_hip_error((int)e); (void)hipFree(base); delete h; return TISE_ERR_HIP; }
    double* p = base;
    h->lt = p; p += dd;
    h->t1 = p; p += dd;
    h->m = p; p += dd;
    h->partial = p; p += nchunk * d;
    h->diag = p; p += d;
    h->colbuf = p; p += d;
    h->va = p; p += d;
    h->vb = p; p += d;
    h->w = p; p += d;
    h->td = p; p += d;
    h->te = p; p += d;
    h->eig = p; p += d;
    h->fz = p; p += fz;
    p += 8;
    h->st = reinterpret_cast<FrState*>(p); p += (sizeof(FrState) + 7) / 8 + 8;
    h->chosen = reinterpret_cast<int*>(p);
    for (int i = 0; i < 8; ++i) {
        e = hipEventCreate(i < 6 ? &h->ev[i] : &h->evp[i - 6]);
        if (e != hipSuccess) { tise_set_last_hip_error((int)e); (void)hipFree(base); delete h; return TISE_ERR_HIP; }
    }
    *out = h;
    return TISE_OK;
}

int tise_frechet_destroy(tise_frechet_t* h) {
    if (!h) return TISE_OK;
    for (int i = 0; i < 6; ++i) (void)hipEventDestroy(h->ev[i]);
    for (int i = 0; i < 2; ++i) (void)hipEventDestroy(h->evp[i]);
    (void)hipFree(h->lt);   // lt is the base of the single allocation
    delete h;
    return TISE_OK;
}

int tise_pivoted_cholesky(tise_frechet_t* h, const double* sigma_dev, double* lt_dev, int* rank_host, void* stream) {
    if (!h || !sigma_dev || !lt_dev || !rank_host) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    const int d = h->d;
    int r = 0;
    h->prefactored = 0;                                        // h->lt is overwritten: a factor left by tise_frechet_prefactor is gone
    int rc = run_pchol(h, sigma_dev, 0.0, &r, st);
    if (rc != TISE_OK) return rc;
    if (r < d) {
        hipLaunchKernelGGL(zero_rows_kernel, dim3(256), dim3(256), 0, st, h->lt, d, r);
        TISE_LAUNCH_CHECK();
    }
    TISE_HIP_CHECK(hipMemcpyAsync(lt_dev, h->lt, (size_t)d * d * sizeof(double), hipMemcpyDeviceToDevice, st));
    TISE_HIP_CHECK(hipStreamSynchronize(st));
    *rank_host = r;
    return TISE_OK;
}

int tise_eigvalsh(tise_frechet_t* h, const double* a_dev, int n, double* w_dev, void* stream) {
    if (!h || !a_dev || !w_dev || n <= 0 || n > h->d) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    TISE_HIP_CHECK(hipMemcpyAsync(h->m, a_dev, (size_t)n * n * sizeof(double), hipMemcpyDeviceToDevice, st));
    int rc = run_eigvalsh_inplace(h, n, st);
    if (rc != TISE_OK) return rc;
    TISE_HIP_CHECK(hipMemcpyAsync(w_dev, h->eig, (size_t)n * sizeof(double), hipMemcpyDeviceToDevice, st));
    return TISE_OK;
}

static int frechet_core(tise_frechet_t* h, const double* mu1_dev, const double* sigma1_dev, const double* mu2_dev,
                        const double* sigma2_dev, double diag_offset, double* out_dev, hipStream_t st, bool factor_here) {
    const int d = h->d;
    int r = 0;
    if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[0], st));
    if (factor_here) {
        h->prefactored = 0;
        int rc = run_chol(h, sigma1_dev, diag_offset, &r, st);
        if (rc != TISE_OK) return rc;
    } else {
        r = h->pf_rank;
    }
    h->last_rank = r;
    if (h->profiling) {
        TISE_HIP_CHECK(hipEventRecord(h->ev[1], st));
        if (r <= 1) { TISE_HIP_CHECK(hipEventRecord(h->ev[2], st)); TISE_HIP_CHECK(hipEventRecord(h->ev[3], st)); }
    }
    int rc = TISE_OK;
    if (r > 0) {
        // T1^T (r x d) = L^T (r x d) * S2 (d x d)
        const bool tri = h->lt_tri && r == d && getenv("TISE_FRECHET_FULL_GEMM") == nullptr;   // A/B switch: the full products
        if (tri) hipLaunchKernelGGL(gemm_f64_tri_kernel<0>, dim3(ceil_div(d, GT_BN), ceil_div(r, GT_BM)), dim3(256), 0, st, h->lt,
                                    (int64_t)d, (int64_t)1, sigma2_dev, (int64_t)d, (int64_t)1, h->t1, (int64_t)d, r, d, d);
        else rc = launch_gemm(h->lt, d, 1, sigma2_dev, d, 1, h->t1, d, r, d, d, st);
        if (rc != TISE_OK) return rc;
        TISE_LAUNCH_CHECK();
        if (diag_offset != 0.0) {   // (S2 + off I): T1^T += off * L^T
            hipLaunchKernelGGL(axpy_kernel, dim3(1024), dim3(256), 0, st, h->t1, h->lt, diag_offset, (int64_t)r * d);
            TISE_LAUNCH_CHECK();
        }
        // M (r x r) = L^T * T1 :  M[a][b] = sum_i LT[a][i] * T1T[b][i]
        if (tri) {
            hipLaunchKernelGGL(gemm_f64_tri_kernel<1>, dim3(ceil_div(r, GT_BN), ceil_div(r, GT_BM)), dim3(256), 0, st, h->lt,
                               (int64_t)d, (int64_t)1, h->t1, (int64_t)1, (int64_t)d, h->m, (int64_t)r, r, r, d);
            hipLaunchKernelGGL(mirror_upper_kernel, dim3(1024), dim3(256), 0, st, h->m, r);
        } else {
            rc = launch_gemm(h->lt, d, 1, h->t1, 1, d, h->m, r, r, r, d, st);
            if (rc != TISE_OK) return rc;
            hipLaunchKernelGGL(symmetrize_kernel, dim3(1024), dim3(256), 0, st, h->m, r);
        }
        TISE_LAUNCH_CHECK();
        if (h->profiling && r > 1) TISE_HIP_CHECK(hipEventRecord(h->ev[2], st));
        rc = run_eigvalsh_inplace(h, r, st);
        if (rc != TISE_OK) return rc;
    }
    if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[4], st));
    hipLaunchKernelGGL(frechet_finish_kernel, dim3(1), dim3(1024), 0, st, mu1_dev, sigma1_dev, mu2_dev, sigma2_dev, d,
                       diag_offset, h->eig, r, h->st, out_dev);
    TISE_LAUNCH_CHECK();
    if (h->profiling) TISE_HIP_CHECK(hipEventRecord(h->ev[5], st));
    return TISE_OK;
}

int tise_frechet_distance(tise_frechet_t* h, const double* mu1_dev, const double* sigma1_dev, const double* mu2_dev,
                          const double* sigma2_dev, double diag_offset, double* out_dev, void* stream) {
    if (!h || !mu1_dev || !sigma1_dev || !mu2_dev || !sigma2_dev || !out_dev) return TISE_ERR_INVALID_ARG;
    return frechet_core(h, mu1_dev, sigma1_dev, mu2_dev, sigma2_dev, diag_offset, out_dev, (hipStream_t)stream, true);
}

// The factor of ONE covariance does not depend on the other: when one side's statistics are known early (the
// reference .npz of the README recipe, fid_score.py:200-203), its pivoted Cholesky can run on a side stream while
// the other side's images are still going through the network, and the serial tail after the last batch is only
// GEMM + tridiagonalisation + bisection.  Tr sqrtm(S1 S2) is symmetric in (S1, S2), so either side may be factored.
int tise_frechet_prefactor(tise_frechet_t* h, const double* sigma_dev, void* stream) {
    if (!h || !sigma_dev) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    h->prefactored = 0;
    TISE_HIP_CHECK(hipEventRecord(h->evp[0], st));
    int r = 0;
    int rc = run_chol(h, sigma_dev, 0.0, &r, st);           // synchronises `st` (rank read-back), not the device
    if (rc != TISE_OK) return rc;
    TISE_HIP_CHECK(hipEventRecord(h->evp[1], st));
    h->pf_rank = r;
    h->prefactored = 1;
    return TISE_OK;
}

int tise_frechet_distance_prefactored(tise_frechet_t* h, const double* mu_f_dev, const double* sigma_f_dev,
                                      const double* mu_o_dev, const double* sigma_o_dev, double* out_dev, void* stream) {
    if (!h || !mu_f_dev || !sigma_f_dev || !mu_o_dev || !sigma_o_dev || !out_dev) return TISE_ERR_INVALID_ARG;
    if (!h->prefactored) return TISE_ERR_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    TISE_HIP_CHECK(hipStreamWaitEvent(st, h->evp[1], 0));   // the factor was produced on another stream
    return frechet_core(h, mu_f_dev, sigma_f_dev, mu_o_dev, sigma_o_dev, 0.0, out_dev, st, false);
}

int tise_frechet_prefactor_ms(tise_frechet_t* h, double* ms_host) {
    if (!h || !ms_host || !h->prefactored) return TISE_ERR_INVALID_ARG;
    TISE_HIP_CHECK(hipEventSynchronize(h->evp[1]));
    float ms = 0.f;
    TISE_HIP_CHECK(hipEventElapsedTime(&ms, h->evp[0], h->evp[1]));
    *ms_host = (double)ms;
    return TISE_OK;
}

int tise_frechet_set_profiling(tise_frechet_t* h, int on) {
    if (!h) return TISE_ERR_INVALID_ARG;
    h->profiling = on ? 1 : 0;
    return TISE_OK;
}

int tise_frechet_phase_ms(tise_frechet_t* h, double* ms_host, int* rank_host) {
    if (!h || !ms_host || !h->profiling) return TISE_ERR_INVALID_ARG;
    TISE_HIP_CHECK(hipEventSynchronize(h->ev[5]));
    for (int i = 0; i < 5; ++i) {
        float ms = 0.f;
        TISE_HIP_CHECK(hipEventElapsedTime(&ms, h->ev[i], h->ev[i + 1]));
        ms_host[i] = (double)ms;
    }
    if (rank_host) *rank_host = h->last_rank;
    return TISE_OK;
}

int tise_gemm_f64(const double* a_dev, int64_t sam, int64_t sak, const double* b_dev, int64_t sbk, int64_t sbn,
                  double* c_dev, int64_t ldc, int m, int n, int k, void* stream) {
    if (!a_dev || !b_dev || !c_dev || m < 0 || n < 0 || k < 0) return TISE_ERR_INVALID_ARG;
    return launch_gemm(a_dev, sam, sak, b_dev, sbk, sbn, c_dev, ldc, m, n, k, (hipStream_t)stream);
}

}  // extern "C"
