// Micro-benchmark: the split-precision inner loop (12 x ds_read_b128 + 3 MFMAs per product, fp32 accumulate) of a
// 32-pixel x 64-cout wave tile and K-step of 32, written with v_mfma_f32_32x32x16_f16 (12 per K-step) and with
// v_mfma_f32_16x16x32_f16 (24 per K-step): same FLOP, same LDS bytes, same accumulator registers.  Random operands.
// Reports wall time, TFLOP/s (fp16-MFMA flop) and the in-kernel clock (s_memtime / s_memrealtime).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/mfma_shape_probe tools/probes/mfma_shape_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// LDS image: rows of 128 B = [hi x32 | lo x32] halves, chunk-swizzled by (row >> 1) & 7; 128 A rows + 256 B rows, 2 stages
constexpr int ROWS = 128 + 256, STAGES = 2;

// wave tile: PM x 32 pixels, TN x 32 couts; per K-step 4 PM + 4 TN ds_read_b128 and 6 PM TN (32x32x16) or 12 PM TN (16x16x32) MFMAs
template <int SHAPE, int WAVES, int PM, int TN>
__global__ __launch_bounds__(WAVES * 64) void probe(const unsigned char* src, float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < ROWS * STAGES * 8; i += WAVES * 64)
        *reinterpret_cast<uint4*>(lds + i * 16) = *reinterpret_cast<const uint4*>(src + ((size_t)blockIdx.x % 7) * 4096 + i * 16);
    __syncthreads();
    const int wm = (wave & 3) % (4 / PM), wn = (wave >> 2) & 1;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    __builtin_amdgcn_s_waitcnt(0xC07F);
    float s = 0.f;
    if (SHAPE == 32) {
        float16_t am[PM][TN], ac[PM][TN];
        for (int i = 0; i < PM; ++i) for (int t = 0; t < TN; ++t) for (int j = 0; j < 16; ++j) { am[i][t][j] = 0.f; ac[i][t][j] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            const unsigned char* st = lds + (it & 1) * ROWS * 128;
            half8_t ah[PM][2], al[PM][2];
#pragma unroll
            for (int i = 0; i < PM; ++i) {
                const int arow = (wm * PM + i) * 32 + (lane & 31);
                const int a0 = arow * 128 + (((lane >> 5) ^ ((arow >> 1) & 7)) << 4);
                ah[i][0] = *reinterpret_cast<const half8_t*>(st + a0); al[i][0] = *reinterpret_cast<const half8_t*>(st + (a0 ^ 64));
                ah[i][1] = *reinterpret_cast<const half8_t*>(st + (a0 ^ 32)); al[i][1] = *reinterpret_cast<const half8_t*>(st + (a0 ^ 96));
            }
#pragma unroll
            for (int t = 0; t < TN; ++t) {
                const int brow = 128 + ((wn * TN + t) % 8) * 32 + (lane & 31);
                const int b0 = brow * 128 + (((lane >> 5) ^ ((brow >> 1) & 7)) << 4);
                half8_t bh[2], bl[2];
                bh[0] = *reinterpret_cast<const half8_t*>(st + b0); bl[0] = *reinterpret_cast<const half8_t*>(st + (b0 ^ 64));
                bh[1] = *reinterpret_cast<const half8_t*>(st + (b0 ^ 32)); bl[1] = *reinterpret_cast<const half8_t*>(st + (b0 ^ 96));
#pragma unroll
                for (int sl = 0; sl < 2; ++sl)
#pragma unroll
                    for (int i = 0; i < PM; ++i) {
                        ac[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bl[sl], ah[i][sl], ac[i][t], 0, 0, 0);
                        am[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl], ah[i][sl], am[i][t], 0, 0, 0);
                        ac[i][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bh[sl], al[i][sl], ac[i][t], 0, 0, 0);
                    }
            }
        }
        for (int i = 0; i < PM; ++i) for (int t = 0; t < TN; ++t) for (int j = 0; j < 16; ++j) s += am[i][t][j] + ac[i][t][j] * (1.f / 2048.f);
    } else {
        float4_t am[2 * PM][2 * TN], ac[2 * PM][2 * TN];
        for (int i = 0; i < 2 * PM; ++i) for (int t = 0; t < 2 * TN; ++t) for (int j = 0; j < 4; ++j) { am[i][t][j] = 0.f; ac[i][t][j] = 0.f; }
        for (int it = 0; it < iters; ++it) {
            const unsigned char* st = lds + (it & 1) * ROWS * 128;
            half8_t ah[2 * PM], al[2 * PM];
#pragma unroll
            for (int i = 0; i < 2 * PM; ++i) {
                const int arow = wm * PM * 32 + i * 16 + (lane & 15);
                const int a0 = arow * 128 + (((lane >> 4) ^ ((arow >> 1) & 7)) << 4);
                ah[i] = *reinterpret_cast<const half8_t*>(st + a0); al[i] = *reinterpret_cast<const half8_t*>(st + (a0 ^ 64));
            }
#pragma unroll
            for (int t = 0; t < 2 * TN; ++t) {
                const int brow = 128 + ((wn * 2 * TN + t) % 16) * 16 + (lane & 15);
                const int b0 = brow * 128 + (((lane >> 4) ^ ((brow >> 1) & 7)) << 4);
                const half8_t bh = *reinterpret_cast<const half8_t*>(st + b0), bl = *reinterpret_cast<const half8_t*>(st + (b0 ^ 64));
#pragma unroll
                for (int i = 0; i < 2 * PM; ++i) {
                    ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], ac[i][t], 0, 0, 0);
                    am[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], am[i][t], 0, 0, 0);
                    ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], ac[i][t], 0, 0, 0);
                }
            }
        }
        for (int i = 0; i < 2 * PM; ++i) for (int t = 0; t < 2 * TN; ++t) for (int j = 0; j < 4; ++j) s += am[i][t][j] + ac[i][t][j] * (1.f / 2048.f);
    }
    out[(size_t)blockIdx.x * WAVES * 64 + tid] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int SHAPE, int WAVES, int PM, int TN>
static void run(const unsigned char* src, float* out, unsigned long long* stamps, int grid, int iters) {
    const size_t ldsb = (size_t)ROWS * STAGES * 128;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<SHAPE, WAVES, PM, TN>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f, total = 0.f;
    int reps = 0;
    while (total < 2500.f) {                     // >= 2.5 s of back-to-back launches, the last one is reported
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<SHAPE, WAVES, PM, TN>), dim3(grid), dim3(WAVES * 64), ldsb, 0, src, out, stamps, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total += ms; ++reps;
    }
    std::vector<unsigned long long> h(2 * grid);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < grid; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    const double flop = (double)grid * WAVES * iters * 6.0 * PM * TN * 32768.0;
    printf("%dx%dx%d  %d waves/SIMD  wave tile %3d px x %3d couts  LDS reads/32x32x16-equivalent MFMA %.2f  %8.3f ms  %7.1f TFLOP/s (fp16 MFMA)  "
           "in-kernel clock %.3f GHz\n", SHAPE, SHAPE, SHAPE == 32 ? 16 : 32, WAVES / 4, PM * 32, TN * 32, (4.0 * PM + 4.0 * TN) / (6.0 * PM * TN), ms,
           flop / ms * 1e-9, clk[clk.size() / 2]);
    fflush(stdout);
}

int main() {
    const size_t nsrc = 7 * 4096 + (size_t)ROWS * STAGES * 128 + 4096;
    std::vector<_Float16> hsrc(nsrc / 2);
    srand(1);
    for (auto& v : hsrc) v = (_Float16)(((rand() % 20001) - 10000) / 10000.0f * 1.5f);
    unsigned char* src; float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&src, nsrc)); CHECK(hipMalloc(&out, 1024 * 512 * 4)); CHECK(hipMalloc(&stamps, 1024 * 16));
    CHECK(hipMemcpy(src, hsrc.data(), nsrc, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 2; ++rep) {
        run<32, 8, 1, 2>(src, out, stamps, 256, 20000);
        run<16, 8, 1, 2>(src, out, stamps, 256, 20000);
        run<32, 8, 2, 2>(src, out, stamps, 256, 10000);
        run<16, 8, 2, 2>(src, out, stamps, 256, 10000);
        run<32, 8, 1, 4>(src, out, stamps, 256, 10000);
        run<16, 8, 1, 4>(src, out, stamps, 256, 10000);
        run<32, 4, 2, 4>(src, out, stamps, 256, 10000);
        run<16, 4, 2, 4>(src, out, stamps, 256, 10000);
    }
    return 0;
}
