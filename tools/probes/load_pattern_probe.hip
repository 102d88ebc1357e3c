// Micro-benchmark behind profiles/r06ac_conv_areg_form.txt: the rate of 16-byte-per-lane global loads out of L2 for the two
// lane -> address maps a convolution's pixel operand can use, per CU:
//   map 0  "DMA piece":   lane i reads chunk (i & 7) of line (i >> 3)          -- 8 lanes per 128-byte line, 8 lines per instruction
//                          (what the LDS-DMA of the default kernel issues; here as plain loads into registers AND as global_load_lds)
//   map 1  "fragment":    lane i reads chunk (i >> 4) of line (i & 15), then the chunk 64 bytes further
//                          -- the MFMA fragment layout: 16 different lines in every 16 consecutive lanes
// Every wave walks its own 32-line window again and again (L1 / L2 resident: the address path is what is measured).
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/load_pattern_probe tools/probes/load_pattern_probe.hip && /tmp/load_pattern_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// MODE 0: map 0 into registers; 1: map 1 into registers; 2: map 0 through global_load_lds (16 bytes per lane)
template <int MODE>
__global__ __launch_bounds__(256) void probe(const unsigned char* __restrict__ src, unsigned* __restrict__ out, int iters, int span_lines) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[4 * 4 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t wbase = ((size_t)blockIdx.x * 4 + wave) * (size_t)span_lines * 128;
    u32x4_t acc = {0u, 0u, 0u, 0u};
    for (int it = 0; it < iters; ++it) {
        const int l0 = (it * 32) % span_lines;                          // 32 lines = 4 KB per iteration and wave, four instructions
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            size_t off;
            if (MODE == 1) off = wbase + (size_t)(l0 + (j >> 1) * 16 + (lane & 15)) * 128 + (lane >> 4) * 16 + (j & 1) * 64;
            else off = wbase + (size_t)(l0 + j * 8 + (lane >> 3)) * 128 + (lane & 7) * 16;
            if (MODE == 2) {
                __builtin_amdgcn_global_load_lds(src + off, (lds_ptr_t)(lds + (wave * 4 + j) * 1024), 16, 0, 0);
            } else {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(src + off);
                acc ^= v;
            }
        }
        if (MODE == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    }
    if (MODE == 2) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = *reinterpret_cast<const u32x4_t*>(lds + threadIdx.x * 16);
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount, wgs = cus * 2, iters = 4096, span = 64;       // 2 workgroups per CU, 8 KB window per wave
    const size_t bytes = (size_t)wgs * 4 * span * 128;
    unsigned char* src; unsigned* out;
    CHECK(hipMalloc((void**)&src, bytes)); CHECK(hipMemset(src, 1, bytes));
    CHECK(hipMalloc((void**)&out, (size_t)wgs * 256 * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const char* names[3] = {"map 0 (8 lanes per line) -> registers", "map 1 (MFMA fragment: 16 lines per 16 lanes) -> registers", "map 0 -> LDS (global_load_lds)"};
    for (int rep = 0; rep < 2; ++rep)
        for (int mode = 0; mode < 3; ++mode) {
            CHECK(hipEventRecord(e0));
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(wgs), dim3(256), 0, 0, src, out, iters, span);
            else if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(wgs), dim3(256), 0, 0, src, out, iters, span);
            else hipLaunchKernelGGL(probe<2>, dim3(wgs), dim3(256), 0, 0, src, out, iters, span);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms = 0.f; CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double total = (double)wgs * 4 * iters * 4096.0;
            if (rep) printf("%-62s %7.3f ms  %8.1f GB/s  %6.1f GB/s per CU  (%.1f B/clk/CU at 2.4 GHz)\n", names[mode], ms, total / ms * 1e-6,
                            total / ms * 1e-6 / cus, total / ms * 1e-6 / cus / 2.4);
        }
    return 0;
}
