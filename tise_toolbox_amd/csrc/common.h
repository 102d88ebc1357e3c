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
