// Shared helpers for libtise_hip.so (gfx950 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/tise_hip.h"

extern "C" void tise_set_last_hip_error(int e);

#define TISE_HIP_CHECK(expr)                          \
    do {                                              \
        hipError_t _e = (expr);                       \
        if (_e != hipSuccess) {                       \
            tise_set_last_hip_error((int)_e);         \
            return TISE_ERR_HIP;                      \
        }                                             \
    } while (0)

#define TISE_LAUNCH_CHECK() TISE_HIP_CHECK(hipGetLastError())

typedef double double4_t __attribute__((ext_vector_type(4)));

// XOR swizzle of the 16-byte chunk index inside a 128-byte LDS row of the convolution kernels' operand images (the DMA
// applies it on the source address, the fragment reads on the LDS address): see conv_split.hip, default kernel.
__host__ __device__ __forceinline__ int tise_lds_swz(int row) { return ((row >> 1) & 3) << 1; }

// "done once per device" latch for per-kernel attributes (hipFuncSetAttribute applies to the CURRENT device): one bit
// per device ordinal, so a process that drives several GPUs sets the attribute on each of them.
#include <atomic>
static inline bool tise_first_use_on_this_device(std::atomic<unsigned long long>& mask) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return true;
    const unsigned long long bit = 1ull << dev;
    return (mask.fetch_or(bit) & bit) == 0;
}

// Compute units of the CURRENT device, cached per device in atomics (launch paths are called from several host threads:
// fid_score._solve_classes; a process may drive GPUs of different sizes).  256 when the query fails.
static inline int tise_cu_count() {
    static std::atomic<int> cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev > 63) return 256;
    int n = cus[dev].load(std::memory_order_relaxed);
    if (n == 0) {
        hipDeviceProp_t prop;
        n = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
        cus[dev].store(n, std::memory_order_relaxed);
    }
    return n;
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// wave-wide (64 lanes) sum of a double
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

// Layout of a split-fp16 activation tensor (v ~= hi + lo * 2^-11) in HBM: NHWC, and inside a pixel the channels in
// blocks of 32 with the two halves of a block side by side,
//        [hi c0..c31 | lo c0..c31] [hi c32..c63 | lo c32..c63] ...          128 bytes per block,
// followed, when C % 32 == 16, by one 64-byte block [hi x16 | lo x16]; a pixel is 4*C bytes.  A K-step of the
// convolution (32 channels of one filter tap, both halves) is therefore ONE 128-byte line per pixel: the LDS-DMA
// path moves 128-byte rows at 38 B/clk/CU against 30 for 64-byte rows (profiles/r01g_glds_rate_microbench.txt), the
// epilogues write whole lines, and the pooling kernels read both halves of a value from the same line.
// tise_ilv_off: offset in fp16 elements of channel c's hi value inside its pixel; the lo value is tise_ilv_second
// elements further.
__host__ __device__ inline int tise_ilv_off(int c, int C) {
    const int full = C & ~31;
    return c < full ? (c >> 5) * 64 + (c & 31) : 2 * full + (c - full);
}
__host__ __device__ inline int tise_ilv_second(int c, int C) { return c < (C & ~31) ? 32 : 16; }

// Range guard of the split-fp16 activation format (v ~= hi + lo * 2^-11, hi = fp16(v)): a value above the fp16
// range (65504) would become +inf in the hi half and poison every later layer without any visible failure.  Every
// kernel that WRITES split tensors keeps the running maximum of what it converts (one v_max per element; all values
// are post-ReLU, i.e. >= 0) and raises bit 0 of a per-device word when it exceeds the range; the host reads and
// clears the words with tise_split_overflow_check (capi.hip; the Python mirror raises FloatingPointError).
// The library is built without relocatable device code, so each translation unit has its OWN word (unnamed
// namespace) and exports a reader for it with TISE_DEFINE_SPLIT_FLAG_READER.
#define TISE_F16_MAX 65504.0f
namespace {
__device__ int g_tise_split_overflow_tu;
__device__ __forceinline__ void tise_flag_split_overflow(float running_max) {
    if (!(running_max <= TISE_F16_MAX)) atomicOr(&g_tise_split_overflow_tu, 1);     // also catches NaN
}
}  // namespace
#define TISE_DEFINE_SPLIT_FLAG_READER(NAME)                                                                          \
    extern "C" int NAME(int* host_flag, void* stream) {                                                             \
        hipStream_t st_ = (hipStream_t)stream;                                                                      \
        int v_ = 0;                                                                                                 \
        TISE_HIP_CHECK(hipMemcpyFromSymbolAsync(&v_, HIP_SYMBOL(g_tise_split_overflow_tu), sizeof(int), 0,          \
                                                hipMemcpyDeviceToHost, st_));                                       \
        TISE_HIP_CHECK(hipStreamSynchronize(st_));                                                                  \
        if (v_) {                                                                                                   \
            const int z_ = 0;                                                                                       \
            TISE_HIP_CHECK(hipMemcpyToSymbolAsync(HIP_SYMBOL(g_tise_split_overflow_tu), &z_, sizeof(int), 0,        \
                                                  hipMemcpyHostToDevice, st_));                                     \
            TISE_HIP_CHECK(hipStreamSynchronize(st_));                                                              \
        }                                                                                                           \
        *host_flag |= v_;                                                                                           \
        return TISE_OK;                                                                                             \
    }
