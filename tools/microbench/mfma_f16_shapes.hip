// Same fp16 MFMA work issued as v_mfma_f32_32x32x16_f16 vs v_mfma_f32_16x16x32_f16 on random operands held in
// registers (no memory traffic in the loop): what the instruction shape alone is worth under this chip's DVFS.
//   hipcc -O3 --offload-arch=gfx950 mfma_f16_shapes.hip -o mfma_f16_shapes && ./mfma_f16_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef float float4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 2) void k32(const half8_t* __restrict__ src, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    half8_t a[2], b[6];
    for (int i = 0; i < 2; ++i) a[i] = src[(tid * 8 + i) & 65535];
    for (int i = 0; i < 6; ++i) b[i] = src[(tid * 8 + 2 + i) & 65535];
    float16_t acc[6];
    for (int t = 0; t < 6; ++t) for (int j = 0; j < 16; ++j) acc[t][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < 6; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(b[t], a[r & 1], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 6; ++t) for (int j = 0; j < 16; ++j) s += acc[t][j];
    out[tid] = s;
}

__global__ __launch_bounds__(256, 2) void k16(const half8_t* __restrict__ src, float* out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x;
    half8_t a[2], b[6];
    for (int i = 0; i < 2; ++i) a[i] = src[(tid * 8 + i) & 65535];
    for (int i = 0; i < 6; ++i) b[i] = src[(tid * 8 + 2 + i) & 65535];
    float4_t acc[24];                                     // same accumulator footprint: 24 x 4 = 6 x 16 registers
    for (int t = 0; t < 24; ++t) for (int j = 0; j < 4; ++j) acc[t][j] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int t = 0; t < 24; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(b[t % 6], a[(r + t) & 1], acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < 24; ++t) for (int j = 0; j < 4; ++j) s += acc[t][j];
    out[tid] = s;
}

int main() {
    std::vector<_Float16> h(65536 * 8);
    srand(1);
    for (auto& v : h) v = (_Float16)((rand() / (float)RAND_MAX) * 4.f - 2.f);
    half8_t* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipMalloc(&out, 2048 * 256 * 4);
    const int iters = 4000;
    for (int rep = 0; rep < 3; ++rep)
        for (int which = 0; which < 2; ++which) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            for (int l = 0; l < 10; ++l) {
                if (which == 0) hipLaunchKernelGGL(k32, dim3(512), dim3(256), 0, 0, src, out, iters);
                else hipLaunchKernelGGL(k16, dim3(512), dim3(256), 0, 0, src, out, iters);
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per wave per iteration: 18 x 32x32x16 = 72 x 16x16x32 MFMAs = 18 * 32768 flop
            const double flop = 10.0 * 512 * 4 * iters * 18.0 * 2 * 32 * 32 * 16;
            if (rep) printf("%s: %.2f ms  %.0f TFLOP/s fp16 (dense peak 2500)\n", which == 0 ? "32x32x16" : "16x16x32", ms, flop / ms / 1e9);
        }
    return 0;
}
