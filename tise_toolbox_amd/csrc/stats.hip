// Streaming fp64 activation statistics {n, s = sum x, S = sum x x^T} from fp32 feature rows.
//
// Replaces pred.cpu().numpy() -> float64 pred_arr, np.mean(act, 0), np.cov(act, rowvar=False)
// (reference image_realism/FID/fid_score.py:98,113,194-195): feature rows stay in HBM, each
// batch is folded into additive sufficient statistics, and (mu, sigma) fall out at the end:
//     mu = s / n          sigma = (S - s s^T / n) / (n - 1)          (np.cov, ddof = 1)
//
// Buffer layout (one allocation, handed as-is to the RCCL all-reduce):
//     [ S : d*d doubles, row-major, only 64x64 tiles with tile_col >= tile_row are written |
//       s : d doubles | n : 1 double | pad : 1 double ]
//
// Kernels
//   syrk_f32_upper_kernel  S += X^T X.  fp32 rows are widened exactly to fp64 on the way into
//       LDS (every fp32*fp32 product is exact in fp64), contraction on v_mfma_f64_16x16x4_f64.
//       Bound: fp64 MFMA.  Algorithmic work per feature row: d*(d+64) flop (upper tiles only).
//   colsum_f32_kernel      s += sum_rows X, n += rows.  Bound: HBM (reads X once: 4*d B/row).
//   stats_finalize_kernel  mu, sigma (mirrors the upper tiles).  Bound: HBM, 3*8*d*d bytes.
#include <new>
#include "common.h"
#include "gemm_tile.h"

struct tise_stats {
    int d;
    int tiles;        // ceil(d / 64)
    double* buf;      // device: S | s | n | pad
    size_t n_doubles;
    double* scratch;  // device: COLSUM_SLICES x tiles x 64 partial column sums | tiles ticket counters (unsigned)
};

// XCD-aware, bijective remap of a linear block id: blocks that share id % 8 share an XCD (and
// its L2), so give each XCD one contiguous run of the tile list (cdna guide T1, bijective form).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = (xcd < r) ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

__global__ __launch_bounds__(256) void syrk_f32_upper_kernel(const float* __restrict__ X, int64_t ld, int rows,
                                                             int d, int tiles, double* __restrict__ S) {
    __shared__ double lds[GT_LDS_DOUBLES];
    const int nwg = tiles * (tiles + 1) / 2;
    int t = xcd_remap(blockIdx.x, nwg);
    // decode upper-triangular tile (tm <= tn) from the row-major list
    int tm = 0;
    while (t >= tiles - tm) { t -= tiles - tm; ++tm; }
    const int tn = tm + t;
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    // A(m,k) = X[k][m0+m], B(k,n) = X[k][n0+n]
    gemm_tile_64x64<float, float>(X, 1, ld, X, ld, 1, d, d, rows, tm * 64, tn * 64, acc, lds);
    gemm_tile_store<1>(S, d, d, d, tm * 64, tn * 64, acc);
}

// Fast path (d % 64 == 0, 16-byte aligned rows): K is walked 64 feature rows at a time.
// The first version (64x64x16 slabs, one slab of loads in flight per workgroup) measured 19 TFLOP/s:
// 3.1 us per 16-row step against 0.43 us of MFMA work -- latency bound (Little's law: ~16 KB in flight
// per CU against a ~2.5 us L2/fabric round trip).  Here a workgroup keeps a 2 x 16 KB slab in flight
// (float4 loads, 8 per thread) under 64 MFMAs per wave (4096 cycles) of work on the previous slab; the
// slab stays fp32 in LDS (pitch 80 floats: the two k-rows a 32-lane half reads land on disjoint banks)
// and is widened to fp64 (exact) in registers.
// COLSUM: the workgroups of the DIAGONAL tiles (tm == tn) also fold the column sums s += sum_rows X of their 64
// columns -- the slab is in LDS anyway -- and tile (0, 0) adds the row count: np.mean's sufficient statistic without a
// second pass over X and without a second launch (the stand-alone colsum kernel was 32 workgroups on 256 CUs, 65-77 us
// per 1000 x 2048 batch = 0.11 TB/s).  Fixed order: thread (column c, phase p) adds rows p, p+4, ... of every slab.
#define SYF_P 80
template <bool COLSUM>
__device__ __forceinline__ void syrk_f32_upper_bk64_body(float* lds, const float* __restrict__ X, int64_t ld, int rows,
                                                         int d, int tiles, double* __restrict__ S,
                                                         double* __restrict__ s, double* __restrict__ n) {
    float* As = lds;
    float* Bs = lds + 64 * SYF_P;
    const int nwg = tiles * (tiles + 1) / 2;
    int t = xcd_remap(blockIdx.x, nwg);
    int tm = 0;
    while (t >= tiles - tm) { t -= tiles - tm; ++tm; }
    const int tn = tm + t;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1, fi = lane & 15, fk = lane >> 4;
    const int lrow = tid >> 4, lcol = (tid & 15) * 4;
    const float* xa = X + tm * 64 + lcol;
    const float* xb = X + tn * 64 + lcol;
    float4 ra[4], rb[4];

// rows past the end are fetched from the last valid row and zeroed by value (a pointer select would
// force the registers into scratch)
#define SYF_FETCH(K0)                                                                              \
    _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                \
        const int r_ = (K0) + lrow + 16 * q;                                                       \
        const float m_ = r_ < rows ? 1.f : 0.f;                                                    \
        const int64_t o_ = (int64_t)(r_ < rows ? r_ : rows - 1) * ld;                              \
        float4 va_ = *reinterpret_cast<const float4*>(xa + o_);                                    \
        float4 vb_ = *reinterpret_cast<const float4*>(xb + o_);                                    \
        ra[q] = make_float4(va_.x * m_, va_.y * m_, va_.z * m_, va_.w * m_);                       \
        rb[q] = make_float4(vb_.x * m_, vb_.y * m_, vb_.z * m_, vb_.w * m_);                       \
    }
    double4_t acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (double4_t){0.0, 0.0, 0.0, 0.0};
    SYF_FETCH(0)
    const float* Bp = Bs;
    const bool diag = COLSUM && tm == tn;                     // workgroup-uniform
    double csum = 0.0;
    for (int k0 = 0; k0 < rows; k0 += 64) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *reinterpret_cast<float4*>(As + (lrow + 16 * q) * SYF_P + lcol) = ra[q];
            *reinterpret_cast<float4*>(Bs + (lrow + 16 * q) * SYF_P + lcol) = rb[q];
        }
        __syncthreads();
        if (k0 + 64 < rows) { SYF_FETCH(k0 + 64) }
        if (diag) {                                           // rows past the end were zeroed by value
#pragma unroll
            for (int kk = 0; kk < 64; kk += 4) csum += (double)As[(kk + wave) * SYF_P + lane];
        }
#pragma unroll
        for (int kk = 0; kk < 64; kk += 4) {
            const double a0 = (double)As[(kk + fk) * SYF_P + wr * 32 + fi];
            const double a1 = (double)As[(kk + fk) * SYF_P + wr * 32 + 16 + fi];
            const double b0 = (double)Bp[(kk + fk) * SYF_P + wc * 32 + fi];
            const double b1 = (double)Bp[(kk + fk) * SYF_P + wc * 32 + 16 + fi];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
        __syncthreads();
    }
    gemm_tile_store<1>(S, d, d, d, tm * 64, tn * 64, acc);
    if (diag) {
        double* part = reinterpret_cast<double*>(lds);        // [4][64]; every wave is past the loop's last barrier
        part[wave * 64 + lane] = csum;
        __syncthreads();
        if (wave == 0) s[tm * 64 + lane] += ((part[lane] + part[64 + lane]) + part[128 + lane]) + part[192 + lane];
        if (tm == 0 && tid == 0) *n += (double)rows;
    }
}

template <bool COLSUM>
__global__ __launch_bounds__(256, 2) void syrk_f32_upper_bk64_kernel(const float* __restrict__ X, int64_t ld, int rows,
                                                                     int d, int tiles, double* __restrict__ S,
                                                                     double* __restrict__ s, double* __restrict__ n) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * SYF_P];
    syrk_f32_upper_bk64_body<COLSUM>(lds, X, ld, rows, d, tiles, S, s, n);
}

// GROUPED form (per-class O-FID, BASELINE configs[4]): the rows of X are sorted by group, group g = rows
// [row0[g], row0[g] + rows[g]) and accumulates into ITS OWN statistics buffer; blockIdx.y = group, blockIdx.x = tile.  One
// launch folds a whole directory's features into all 80 classes' {n, s, S} -- each S tile is read-modify-written ONCE per
// directory (round 4: one index_select + one 528-workgroup launch per class and DEVICE BATCH, ~5.4 GB of accumulator traffic
// per batch for 8 MB of features).  The table travels as a kernel argument (2 KB).
#define SYRK_MAX_GROUPS 128
struct SyrkGroups {
    double* buf[SYRK_MAX_GROUPS];
    int row0[SYRK_MAX_GROUPS];
    int rows[SYRK_MAX_GROUPS];
};
__global__ __launch_bounds__(256, 2) void syrk_f32_upper_bk64_grouped_kernel(const float* __restrict__ X, int64_t ld, int d,
                                                                             int tiles, const SyrkGroups g) {
    __shared__ __attribute__((aligned(16))) float lds[2 * 64 * SYF_P];
    const int grp = blockIdx.y;
    const int rows = g.rows[grp];
    if (rows <= 0) return;                                    // workgroup-uniform
    double* buf = g.buf[grp];
    double* s = buf + (size_t)d * d;
    syrk_f32_upper_bk64_body<true>(lds, X + (int64_t)g.row0[grp] * ld, ld, rows, d, tiles, buf, s, s + d);
}

// Stand-alone column sums (shapes the fused kernel does not serve, and tise_stats_update_sum).  Grid = column tiles x
// row slices, so d = 2048 launches 32 x 8 = 256 workgroups; a slice's partial sums go to scratch and the LAST
// workgroup of a column tile to arrive (ticket counter) adds the slices in slice order: fixed order => reproducible.
#define COLSUM_SLICES 8
__global__ __launch_bounds__(256) void colsum_f32_sliced_kernel(const float* __restrict__ X, int64_t ld, int rows, int d,
                                                                double* __restrict__ s, double* __restrict__ n,
                                                                double* __restrict__ scratch, unsigned* __restrict__ tickets) {
    __shared__ double part[4][64];
    __shared__ unsigned s_last;
    const int lc = threadIdx.x & 63, ph = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lc;
    const int per = (rows + COLSUM_SLICES - 1) / COLSUM_SLICES;
    const int r0 = blockIdx.y * per, r1 = min(rows, r0 + per);
    double acc = 0.0;
    if (c < d)
        for (int r = r0 + ph; r < r1; r += 4) acc += (double)X[(int64_t)r * ld + c];
    part[ph][lc] = acc;
    __syncthreads();
    if (ph == 0) scratch[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 64 + lc] = ((part[0][lc] + part[1][lc]) + part[2][lc]) + part[3][lc];
    __threadfence();                                          // the slice's partials are visible device-wide before the ticket
    __syncthreads();
    if (threadIdx.x == 0) s_last = atomicAdd(&tickets[blockIdx.x], 1u);
    __syncthreads();
    if (s_last != COLSUM_SLICES - 1) return;
    __threadfence();
    if (ph == 0 && c < d) {
        double t = 0.0;
        for (int y = 0; y < COLSUM_SLICES; ++y)
            t += scratch[((int64_t)y * gridDim.x + blockIdx.x) * 64 + lc];
        s[c] += t;
    }
    if (threadIdx.x == 0) {
        tickets[blockIdx.x] = 0;                              // ready for the next launch (stream order)
        if (blockIdx.x == 0) *n += (double)rows;
    }
}

// 256 threads = 64 columns x 4 row phases; fixed summation order => bitwise reproducible.
__global__ __launch_bounds__(256) void colsum_f32_kernel(const float* __restrict__ X, int64_t ld, int rows, int d,
                                                         double* __restrict__ s, double* __restrict__ n) {
    __shared__ double part[4][64];
    const int c = blockIdx.x * 64 + (threadIdx.x & 63);
    const int ph = threadIdx.x >> 6;
    double acc = 0.0;
    if (c < d)
        for (int r = ph; r < rows; r += 4) acc += (double)X[(int64_t)r * ld + c];
    part[ph][threadIdx.x & 63] = acc;
    __syncthreads();
    if (ph == 0 && c < d) s[c] += ((part[0][threadIdx.x] + part[1][threadIdx.x]) + part[2][threadIdx.x]) + part[3][threadIdx.x];
    if (blockIdx.x == 0 && threadIdx.x == 0) *n += (double)rows;
}

__global__ __launch_bounds__(256) void stats_finalize_kernel(const double* __restrict__ S, const double* __restrict__ s,
                                                             const double* __restrict__ n, int d,
                                                             double* __restrict__ mu, double* __restrict__ sigma) {
    const double nn = *n;
    const int64_t total = (int64_t)d * d;
    for (int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(e / d), j = (int)(e % d);
        // S holds tile (ti, tj) only for tj >= ti; inside a diagonal tile the full 64x64 block is present
        const int ti = i >> 6, tj = j >> 6;
        const double sij = (tj >= ti) ? S[(int64_t)i * d + j] : S[(int64_t)j * d + i];
        sigma[e] = (sij - s[i] * s[j] / nn) / (nn - 1.0);
        if (j == 0) mu[i] = s[i] / nn;
    }
}

extern "C" {

int tise_stats_create(int d, tise_stats_t** out) {
    if (d <= 0 || d > 8192 || out == nullptr) return TISE_ERR_INVALID_ARG;
    tise_stats* h = new (std::nothrow) tise_stats;
    if (!h) return TISE_ERR_INVALID_ARG;
    h->d = d;
    h->tiles = ceil_div(d, 64);
    h->n_doubles = (size_t)d * d + d + 2;
    h->buf = nullptr;
    hipError_t e = hipMalloc((void**)&h->buf, h->n_doubles * sizeof(double));
    if (e != hipSuccess) { tise_set_last_hip_error((int)e); delete h; return TISE_ERR_HIP; }
    e = hipMemset(h->buf, 0, h->n_doubles * sizeof(double));
    if (e != hipSuccess) { tise_set_last_hip_error((int)e); hipFree(h->buf); delete h; return TISE_ERR_HIP; }
    const size_t scratch_bytes = ((size_t)8 * h->tiles * 64) * sizeof(double) + (size_t)h->tiles * sizeof(unsigned) + 64;
    h->scratch = nullptr;
    e = hipMalloc((void**)&h->scratch, scratch_bytes);
    if (e == hipSuccess) e = hipMemset(h->scratch, 0, scratch_bytes);
    if (e != hipSuccess) { tise_set_last_hip_error((int)e); hipFree(h->buf); if (h->scratch) hipFree(h->scratch); delete h; return TISE_ERR_HIP; }
    *out = h;
    return TISE_OK;
}

int tise_stats_destroy(tise_stats_t* h) {
    if (!h) return TISE_OK;
    hipFree(h->buf);
    hipFree(h->scratch);
    delete h;
    return TISE_OK;
}

int tise_stats_reset(tise_stats_t* h, void* stream) {
    if (!h) return TISE_ERR_INVALID_ARG;
    TISE_HIP_CHECK(hipMemsetAsync(h->buf, 0, h->n_doubles * sizeof(double), (hipStream_t)stream));
    return TISE_OK;
}

static int stats_check(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld) {
    if (!h || (!feats_dev && rows > 0) || rows < 0 || ld < h->d || rows > (int64_t)1 << 30) return TISE_ERR_INVALID_ARG;
    return TISE_OK;
}

int tise_stats_update_cov(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream) {
    int rc = stats_check(h, feats_dev, rows, ld);
    if (rc != TISE_OK || rows == 0) return rc;
    const int nwg = h->tiles * (h->tiles + 1) / 2;
    const bool fast = (h->d % 64 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(feats_dev) & 15) == 0);
    if (fast)
        hipLaunchKernelGGL(syrk_f32_upper_bk64_kernel<false>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, feats_dev, ld,
                           (int)rows, h->d, h->tiles, h->buf, (double*)nullptr, (double*)nullptr);
    else
        hipLaunchKernelGGL(syrk_f32_upper_kernel, dim3(nwg), dim3(256), 0, (hipStream_t)stream, feats_dev, ld, (int)rows,
                           h->d, h->tiles, h->buf);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_stats_update_sum(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream) {
    int rc = stats_check(h, feats_dev, rows, ld);
    if (rc != TISE_OK || rows == 0) return rc;
    double* s = h->buf + (size_t)h->d * h->d;
    if (rows >= 64 * COLSUM_SLICES) {
        unsigned* tickets = reinterpret_cast<unsigned*>(h->scratch + (size_t)COLSUM_SLICES * h->tiles * 64);
        hipLaunchKernelGGL(colsum_f32_sliced_kernel, dim3(h->tiles, COLSUM_SLICES), dim3(256), 0, (hipStream_t)stream,
                           feats_dev, ld, (int)rows, h->d, s, s + h->d, h->scratch, tickets);
    } else {
        hipLaunchKernelGGL(colsum_f32_kernel, dim3(h->tiles), dim3(256), 0, (hipStream_t)stream, feats_dev, ld, (int)rows,
                           h->d, s, s + h->d);
    }
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

int tise_stats_update(tise_stats_t* h, const float* feats_dev, int64_t rows, int64_t ld, void* stream) {
    int rc = stats_check(h, feats_dev, rows, ld);
    if (rc != TISE_OK || rows == 0) return rc;
    const bool fast = (h->d % 64 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(feats_dev) & 15) == 0);
    if (fast) {                                               // ONE launch: covariance tiles + column sums + row count
        const int nwg = h->tiles * (h->tiles + 1) / 2;
        double* s = h->buf + (size_t)h->d * h->d;
        hipLaunchKernelGGL(syrk_f32_upper_bk64_kernel<true>, dim3(nwg), dim3(256), 0, (hipStream_t)stream, feats_dev, ld,
                           (int)rows, h->d, h->tiles, h->buf, s, s + h->d);
        TISE_LAUNCH_CHECK();
        return TISE_OK;
    }
    rc = tise_stats_update_cov(h, feats_dev, rows, ld, stream);
    if (rc != TISE_OK) return rc;
    return tise_stats_update_sum(h, feats_dev, rows, ld, stream);
}

int tise_stats_update_grouped(tise_stats_t* const* handles, int n_groups, const float* feats_dev, const int64_t* row_offsets,
                              int64_t ld, void* stream) {
    if (!handles || n_groups < 0 || !row_offsets || (!feats_dev && n_groups > 0 && row_offsets[n_groups] > 0)) return TISE_ERR_INVALID_ARG;
    if (n_groups == 0) return TISE_OK;
    const int d = handles[0] ? handles[0]->d : 0;
    for (int g = 0; g < n_groups; ++g) {
        if (!handles[g] || handles[g]->d != d || row_offsets[g + 1] < row_offsets[g] || row_offsets[g] < 0 ||
            row_offsets[g + 1] > ((int64_t)1 << 30))
            return TISE_ERR_INVALID_ARG;
    }
    if (ld < d) return TISE_ERR_INVALID_ARG;
    const bool fast = (d % 64 == 0) && (ld % 4 == 0) && ((reinterpret_cast<uintptr_t>(feats_dev) & 15) == 0);
    if (!fast) {                                              // shapes the tiled kernel does not serve: group by group
        for (int g = 0; g < n_groups; ++g) {
            const int rc = tise_stats_update(handles[g], feats_dev + row_offsets[g] * ld, row_offsets[g + 1] - row_offsets[g], ld, stream);
            if (rc != TISE_OK) return rc;
        }
        return TISE_OK;
    }
    const int tiles = handles[0]->tiles;
    const int nwg = tiles * (tiles + 1) / 2;
    for (int g0 = 0; g0 < n_groups; g0 += SYRK_MAX_GROUPS) {
        const int ng = n_groups - g0 < SYRK_MAX_GROUPS ? n_groups - g0 : SYRK_MAX_GROUPS;
        SyrkGroups tab;
        for (int g = 0; g < SYRK_MAX_GROUPS; ++g) {
            const bool on = g < ng;
            tab.buf[g] = on ? handles[g0 + g]->buf : nullptr;
            tab.row0[g] = on ? (int)row_offsets[g0 + g] : 0;
            tab.rows[g] = on ? (int)(row_offsets[g0 + g + 1] - row_offsets[g0 + g]) : 0;
        }
        hipLaunchKernelGGL(syrk_f32_upper_bk64_grouped_kernel, dim3(nwg, ng), dim3(256), 0, (hipStream_t)stream, feats_dev, ld, d,
                           tiles, tab);
        TISE_LAUNCH_CHECK();
    }
    return TISE_OK;
}

int tise_stats_buffer(tise_stats_t* h, double** buf_dev, size_t* n_doubles) {
    if (!h || !buf_dev || !n_doubles) return TISE_ERR_INVALID_ARG;
    *buf_dev = h->buf;
    *n_doubles = h->n_doubles;
    return TISE_OK;
}

int tise_stats_finalize(tise_stats_t* h, double* mu_dev, double* sigma_dev, void* stream) {
    if (!h || !mu_dev || !sigma_dev) return TISE_ERR_INVALID_ARG;
    const int d = h->d;
    const double* S = h->buf;
    const double* s = h->buf + (size_t)d * d;
    const double* n = s + d;
    int blocks = (int)((((int64_t)d * d) + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(stats_finalize_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, S, s, n, d, mu_dev, sigma_dev);
    TISE_LAUNCH_CHECK();
    return TISE_OK;
}

}  // extern "C"
