// Micro-benchmark (round 5): does ENFORCED anti-phase between the two waves of a SIMD raise the matrix pipe's duty cycle?
//
// The default convolution kernel (csrc/conv_split.hip) is 256 threads, two workgroups per CU: the two waves of a SIMD belong
// to different workgroups, each K-step is  s_waitcnt vmcnt(0) -> s_barrier -> issue the NEXT step's 8 LDS-DMA instructions ->
// fragment reads + 48 MFMAs (64 px x 64 couts per wave, three 16x16x32 MFMAs per product).  DESIGN.md section 4a measured
// DMA-only and MFMA-only launches at ~60-70 % of a full launch each ("the two hardly overlap"); MI355X_MICROARCH.md ("Two waves
// per SIMD", item 9) reports 4-8 % from staggering the halves of a 512-thread workgroup by half a block.
//
// This probe runs that K-step (same LDS image, same fragment reads, same MFMA sequence, real global_load_lds traffic from an
// L2-resident source) in three forms:
//   A  two independent 256-thread workgroups per CU                          (today's structure)
//   B  one 512-thread workgroup = two halves with their own tiles and LDS regions, TWO joint barriers per K-step, half 1 one
//      barrier behind: in every barrier interval one half issues its DMA while the other runs its MFMAs
//   C  as B without the offset (both halves in phase): the control
//   D  as B with the phase barrier as a bare s_barrier (no s_waitcnt vmcnt(0) in front of it)
// and prints TFLOP/s (fp16 MFMA) and the in-kernel clock.  hipcc -O3 --offload-arch=gfx950 -o /tmp/kstep_phase_probe kstep_phase_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int STAGE = 32 * 1024;                 // 128 pixel rows + 128 cout rows of 128 B
constexpr int REGION = 2 * STAGE;                // two stages per tile
#ifndef NSRC
#define NSRC 96                                  // distinct 2 MB source streams: 96 = 192 MB (Infinity Cache), 2 = 4 MB (L2)
#endif
constexpr int SRC_PER_TILE = 64 * STAGE;         // 2 MB of source per tile stream, walked cyclically

__device__ __forceinline__ int swz(int row) { return ((row >> 1) & 3) << 1; }

// one K-step of a half (4 waves, 2 x 2, 64 px x 64 couts each): fragment reads + 48 MFMAs on stage `st`
__device__ __forceinline__ void compute(const unsigned char* st, int lane, int wm, int wn, float4_t (&am)[4][4], float4_t (&ac)[4][4]) {
    const int f16o = (lane & 15) * 128 + (((lane >> 4) ^ swz(lane & 15)) << 4);
    half8_t ah[4], al[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned char* p = st + (wm * 4 + i) * 2048 + f16o;
        ah[i] = *reinterpret_cast<const half8_t*>(p);
        al[i] = *reinterpret_cast<const half8_t*>(st + (wm * 4 + i) * 2048 + (f16o ^ 64));
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const unsigned char* b = st + 128 * 128 + (wn * 4 + t) * 2048;
        const half8_t bh = *reinterpret_cast<const half8_t*>(b + f16o), bl = *reinterpret_cast<const half8_t*>(b + (f16o ^ 64));
#ifndef ORDER
#define ORDER 0
#endif
#if ORDER == 0
        // the product kernel's order: the two MFMAs on the correction accumulator of a block are one MFMA apart
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], ac[i][t], 0, 0, 0);
            am[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], am[i][t], 0, 0, 0);
            ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], ac[i][t], 0, 0, 0);
        }
#else
        // ORDER 1: every accumulator is touched again only after >= 7 other MFMAs
#pragma unroll
        for (int i = 0; i < 4; ++i) ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bl, ah[i], ac[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) am[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, ah[i], am[i][t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) ac[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(bh, al[i], ac[i][t], 0, 0, 0);
#endif
    }
}

// the step's DMA: 8 pieces of 1 KB per wave (4 pixel pieces + 4 weight pieces) into stage `dst`
__device__ __forceinline__ void issue(const unsigned char* src, unsigned char* dst, int wave4, int lane) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const unsigned char* s = src + (wave4 * 8 + j) * 1024 + lane * 16;
        __builtin_amdgcn_global_load_lds(s, (lds_ptr_t)(dst + (wave4 * 8 + j) * 1024), 16, 0, 0);
    }
}

// MODE 0: 256-thread workgroup, one tile (launch two per CU).  MODE 1: 512 threads, two halves in anti-phase.  MODE 2: in phase.
template <int MODE>
__global__ __launch_bounds__(MODE == 0 ? 256 : 512) void probe(const unsigned char* src, float* out, unsigned long long* stamps, int iters) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int half = MODE == 0 ? 0 : wave >> 2, wave4 = wave & 3;
    const int wm = wave4 & 1, wn = wave4 >> 1;
    unsigned char* my = lds + half * REGION;
    const int tile = MODE == 0 ? blockIdx.x : 2 * blockIdx.x + half;
    const unsigned char* tsrc = src + (size_t)(tile % NSRC) * SRC_PER_TILE;
    float4_t am[4][4], ac[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t) { am[i][t] = (float4_t){0.f, 0.f, 0.f, 0.f}; ac[i][t] = (float4_t){0.f, 0.f, 0.f, 0.f}; }
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    issue(tsrc, my, wave4, lane);
    if ((MODE == 1 || MODE == 3) && half == 1) __syncthreads();              // half 1 runs one barrier behind
    for (int it = 0; it < iters; ++it) {
        unsigned char* cur = my + (it & 1) * STAGE;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue(tsrc + (size_t)((it + 1) & 63) * STAGE, my + ((it + 1) & 1) * STAGE, wave4, lane);
        if (MODE == 1 || MODE == 2) __syncthreads();           // the partner half's "DMA landed" barrier (hipcc adds s_waitcnt vmcnt(0): this half's DMA lands first)
        if (MODE == 3) {                                        // ... as a bare s_barrier: the DMA just issued stays in flight under the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        compute(cur, lane, wm, wn, am, ac);
    }
    if ((MODE == 1 || MODE == 3) && half == 0) __syncthreads();              // balance the count
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += am[i][t][j] + ac[i][t][j] * (1.f / 2048.f);
    out[(size_t)blockIdx.x * 512 + tid] = s;
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) { stamps[2 * blockIdx.x] = t1 - t0; stamps[2 * blockIdx.x + 1] = r1 - r0; }
}

template <int MODE>
static void run(const char* name, const unsigned char* src, float* out, unsigned long long* stamps, int iters) {
    const int threads = MODE == 0 ? 256 : 512, grid = MODE == 0 ? 512 : 256;
    const size_t ldsb = MODE == 0 ? REGION : 2 * REGION;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsb));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    float ms = 0.f, total = 0.f;
    while (total < 2000.f) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL((probe<MODE>), dim3(grid), dim3(threads), ldsb, 0, src, out, stamps, iters);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        total += ms;
    }
    std::vector<unsigned long long> h(2 * grid);
    CHECK(hipMemcpy(h.data(), stamps, h.size() * 8, hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (int b = 0; b < grid; ++b) clk.push_back((double)h[2 * b] / (double)h[2 * b + 1] * 0.1);
    std::sort(clk.begin(), clk.end());
    const double flop = 512.0 * 4 /*waves per tile*/ * iters * 48.0 * 16384.0;        // 512 tiles in every mode
    printf("%-58s %8.3f ms  %7.1f TFLOP/s (fp16 MFMA)  clock %.3f GHz  %.0f shader clocks per K-step\n", name, ms, flop / ms * 1e-9,
           clk[clk.size() / 2], ms * 1e-3 * clk[clk.size() / 2] * 1e9 / iters);
    fflush(stdout);
}

int main() {
    const size_t nsrc = (size_t)NSRC * SRC_PER_TILE + 4096;
    std::vector<_Float16> hsrc(nsrc / 2);
    srand(1);
    for (auto& v : hsrc) v = (_Float16)(((rand() % 20001) - 10000) / 10000.0f * 1.5f);
    unsigned char* src; float* out; unsigned long long* stamps;
    CHECK(hipMalloc(&src, nsrc)); CHECK(hipMalloc(&out, 1024 * 512 * 4)); CHECK(hipMalloc(&stamps, 1024 * 16));
    CHECK(hipMemcpy(src, hsrc.data(), nsrc, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; ++rep) {
        run<0>("A  two independent 256-thread workgroups per CU", src, out, stamps, 4000);
        run<1>("B  512 threads, halves in ANTI-phase (2 joint barriers / step)", src, out, stamps, 4000);
        run<2>("C  512 threads, halves IN phase (2 joint barriers / step)", src, out, stamps, 4000);
        run<3>("D  as B, phase barrier without the DMA wait (bare s_barrier)", src, out, stamps, 4000);
    }
    return 0;
}
