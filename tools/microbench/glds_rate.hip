// global_load_lds_dwordx4 issue rate per CU for different source address shapes of one 1 KB piece
// (64 lanes x 16 B): how much of the ~64 B/clk/CU vector-memory path the conv kernels' 16-row x 64-B pieces get.
//   hipcc -O3 --offload-arch=gfx950 glds_rate.hip -o glds_rate && ./glds_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((address_space(3))) void* lds_ptr_t;

// pattern: rows of RB bytes (RB = 64, 128, 256, 1024), consecutive rows STRIDE bytes apart
template <int RB>
__global__ __launch_bounds__(512, 1) void k(const unsigned char* __restrict__ src, size_t span, int stride, int iters,
                                            unsigned long long* cyc, float* sink) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = RB / 16;                 // lanes per row
    const int row = lane / LPR, chunk = lane % LPR;
    constexpr int ROWS = 64 / LPR;               // rows per piece
    // every wave walks its own region; pieces advance by ROWS rows
    size_t off = ((size_t)blockIdx.x * 8 + wave) * (size_t)ROWS * stride * 64 % span;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const unsigned char* s = src + (off + (size_t)(u * ROWS + row) * stride + chunk * 16) % span;
            unsigned char* d = lds + wave * 8192 + u * 1024;
            __builtin_amdgcn_global_load_lds(s, (lds_ptr_t)d, 16, 0, 0);
        }
        off += (size_t)8 * ROWS * stride;
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (lds[threadIdx.x] == 123 && sink) sink[0] = 1.f;
}

template <int RB>
__global__ __launch_bounds__(512, 1) void kreg(const unsigned char* __restrict__ src, size_t span, int stride, int iters,
                                               unsigned long long* cyc, float* sink) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int LPR = RB / 16;
    const int row = lane / LPR, chunk = lane % LPR;
    constexpr int ROWS = 64 / LPR;
    size_t off = ((size_t)blockIdx.x * 8 + wave) * (size_t)ROWS * stride * 64 % span;
    typedef unsigned int u4 __attribute__((ext_vector_type(4)));
    u4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
        u4 v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u)
            v[u] = *reinterpret_cast<const u4*>(src + (off + (size_t)(u * ROWS + row) * stride + chunk * 16) % span);
#pragma unroll
        for (int u = 0; u < 8; ++u) acc ^= v[u];
        off += (size_t)8 * ROWS * stride;
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    if (acc[0] + acc[1] + acc[2] + acc[3] == 12345u && sink) sink[0] = 1.f;
}

int main() {
    const size_t span = 48u << 20;               // 48 MB: L2 (8 x 4 MB) + MALL resident
    unsigned char* src; unsigned long long* cyc; 
    hipMalloc(&src, span + (4 << 20)); hipMemset(src, 1, span + (4 << 20));
    hipMalloc(&cyc, 256 * 8);
    const int iters = 400;
    struct P { int rb, stride; const char* name; } ps[] = {
        {1024, 1024, "1 row x 1024 B (contiguous)"}, {256, 256, "4 rows x 256 B contiguous"},
        {128, 128, "8 rows x 128 B contiguous"}, {128, 1152, "8 rows x 128 B, stride 1152"},
        {64, 64, "16 rows x 64 B contiguous"}, {64, 128, "16 rows x 64 B, stride 128"},
        {64, 576, "16 rows x 64 B, stride 576"}, {64, 1536, "16 rows x 64 B, stride 1536"},
        {128, 1536, "8 rows x 128 B, stride 1536"}, {256, 1536, "4 rows x 256 B, stride 1536"}};
    for (auto& p : ps) {
        for (int rep = 0; rep < 4; ++rep) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0);
            void (*fn)(const unsigned char*, size_t, int, int, unsigned long long*, float*) =
                p.rb == 1024 ? k<1024> : p.rb == 256 ? k<256> : p.rb == 128 ? k<128> : k<64>;
            hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
            void (*fr)(const unsigned char*, size_t, int, int, unsigned long long*, float*) =
                p.rb == 1024 ? kreg<1024> : p.rb == 256 ? kreg<256> : p.rb == 128 ? kreg<128> : kreg<64>;
            if (rep < 2) hipLaunchKernelGGL(fn, dim3(256), dim3(512), 65536, 0, src, span, p.stride, iters, cyc, (float*)nullptr);
            else hipLaunchKernelGGL(fr, dim3(256), dim3(512), 0, 0, src, span, p.stride, iters, cyc, (float*)nullptr);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto v : h) avg += v; avg /= 256;
            const double bytes_cu = (double)iters * 8 * 8 * 1024;   // per CU
            if (rep == 1) printf("%-34s %7.3f ms  %6.1f cycles/piece/CU  %5.1f B/clk/CU (s_memtime)  %6.2f TB/s aggregate\n", p.name, ms,
                            avg / (iters * 64.0), bytes_cu / avg, bytes_cu * 256 / (ms * 1e-3) / 1e12);
            if (rep == 3) printf("   plain global_load_dwordx4 -> VGPR   %7.3f ms  %6.1f cycles/piece/CU  %5.1f B/clk/CU  %6.2f TB/s\n", ms,
                            avg / (iters * 64.0), bytes_cu / avg, bytes_cu * 256 / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
