// Library-level entry points of libtise_hip.so: status strings, last HIP error, device probe.
#include <atomic>
#include "common.h"

static std::atomic<int> g_last_hip_error{0};

extern "C" {

void tise_set_last_hip_error(int e) { g_last_hip_error.store(e); }

int tise_last_hip_error(void) { return g_last_hip_error.load(); }

int tise_version(void) { return 1; }

const char* tise_status_string(int status) {
    switch (status) {
        case TISE_OK: return "ok";
        case TISE_ERR_INVALID_ARG: return "invalid argument";
        case TISE_ERR_HIP: return "HIP runtime error";
        case TISE_ERR_NO_DEVICE: return "no gfx950 device";
        case TISE_ERR_UNSUPPORTED: return "unsupported size";
        default: return "unknown status";
    }
}

int tise_device_info(int* cu_count, int* gcn_arch_is_gfx950, size_t* total_mem) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) { tise_set_last_hip_error((int)e); return TISE_ERR_NO_DEVICE; }
    int dev = 0;
    TISE_HIP_CHECK(hipGetDevice(&dev));
    hipDeviceProp_t prop;
    TISE_HIP_CHECK(hipGetDeviceProperties(&prop, dev));
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (gcn_arch_is_gfx950) {
        const char* a = prop.gcnArchName;
        *gcn_arch_is_gfx950 = (a[0] == 'g' && a[1] == 'f' && a[2] == 'x' && a[3] == '9' && a[4] == '5' && a[5] == '0') ? 1 : 0;
    }
    if (total_mem) *total_mem = prop.totalGlobalMem;
    return TISE_OK;
}

// ---- host feed: page-lock caller-owned host memory and copy from it asynchronously ------------------------------------
// (a2) the PNG decode workers of tise_toolbox_amd/png_ring.py write decoded uint8 pixels into ONE shared-memory ring; the
// parent page-locks that ring once and enqueues host->device copies of finished chunks straight from it (no collate, no
// pickling, no per-batch pin_memory) -- the hand-over the reference does through DataLoader worker queues
// (image_realism/FID/fid_score.py:215-217).
int tise_host_register(void* host_ptr, size_t bytes) {
    if (!host_ptr || bytes == 0) return TISE_ERR_INVALID_ARG;
    TISE_HIP_CHECK(hipHostRegister(host_ptr, bytes, hipHostRegisterDefault));
    return TISE_OK;
}

int tise_host_unregister(void* host_ptr) {
    if (!host_ptr) return TISE_ERR_INVALID_ARG;
    TISE_HIP_CHECK(hipHostUnregister(host_ptr));
    return TISE_OK;
}

int tise_memcpy_h2d_async(void* dst_dev, const void* src_host, size_t bytes, void* stream) {
    if ((!dst_dev || !src_host) && bytes > 0) return TISE_ERR_INVALID_ARG;
    if (bytes == 0) return TISE_OK;
    TISE_HIP_CHECK(hipMemcpyAsync(dst_dev, src_host, bytes, hipMemcpyHostToDevice, (hipStream_t)stream));
    return TISE_OK;
}

// per-translation-unit range-guard words (common.h)
int tise_internal_split_flag_conv_split(int* host_flag, void* stream);
int tise_internal_split_flag_conv_pipe(int* host_flag, void* stream);
int tise_internal_split_flag_trunk_ops(int* host_flag, void* stream);

int tise_split_overflow_check(int* flag_host, void* stream) {
    if (!flag_host) return TISE_ERR_INVALID_ARG;
    *flag_host = 0;
    int rc = tise_internal_split_flag_conv_split(flag_host, stream);
    if (rc == TISE_OK) rc = tise_internal_split_flag_conv_pipe(flag_host, stream);
    if (rc == TISE_OK) rc = tise_internal_split_flag_trunk_ops(flag_host, stream);
    return rc;
}

}  // extern "C"
