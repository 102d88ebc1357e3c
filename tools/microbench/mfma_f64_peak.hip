// Issue-rate microbenchmark for v_mfma_f64_16x16x4_f64 on gfx950 (the guide's MFMA table has no f64 row).
// Each wave runs N back-to-back MFMAs on NACC independent accumulators (operands in registers);
// grid = 256 CUs x waves_per_cu.  Prints cycles per MFMA per SIMD and chip TFLOP/s.
// build: hipcc -O3 --offload-arch=gfx950 mfma_f64_peak.hip -o mfma_f64_peak
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef double double4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ void k(double* out, int iters, unsigned long long* cyc) {
    double4_t acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (double4_t){0, 0, 0, 0};
    double a = 1.0 + threadIdx.x * 1e-3, b = 0.5 + threadIdx.x * 1e-4;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    double s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC>
void run(int waves_per_cu, int iters) {
    int cus = 256;
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0); cus = p.multiProcessorCount;
    int threads = 256, blocks = cus * waves_per_cu / 4;
    double* out; unsigned long long* cyc;
    hipMalloc(&out, (size_t)blocks * threads * 8); hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<blocks, threads>>>(out, iters, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<blocks, threads>>>(out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    double n_mfma = (double)blocks * (threads / 64) * iters * NACC;
    double tflops = n_mfma * 2048.0 / (ms * 1e-3) / 1e12;
    double waves_per_simd = waves_per_cu / 4.0;
    printf("NACC=%d waves/CU=%d: %.3f ms, %.1f TFLOP/s, wave0: %.1f cycles per MFMA per wave (=> %.1f cyc/MFMA/SIMD at %.2f waves/SIMD)\n",
           NACC, waves_per_cu, ms, tflops, (double)c / (iters * (double)NACC), (double)c / (iters * (double)NACC) / (waves_per_simd < 1 ? 1 : waves_per_simd), waves_per_simd);
    hipFree(out); hipFree(cyc);
}

int main() {
    run<1>(4, 20000);
    run<4>(4, 5000);
    run<8>(4, 2500);
    run<4>(8, 5000);
    run<8>(16, 2500);
    return 0;
}
