"""ctypes binding of libtise_hip.so (the C ABI declared in include/tise_hip.h).

There is NO CPU fallback: if the shared library is missing or fails to load, every
entry point raises ``TiseLibraryError``.  The numpy oracle under ``oracle/`` is test
infrastructure and is never imported from here.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_double, c_float, c_int, c_int64, c_size_t, c_uint8, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TISE_LIB_PATH") or os.path.join(_HERE, "libtise_hip.so")   # override: kernel A/B builds


class TiseLibraryError(RuntimeError):
    pass


class TiseStatusError(RuntimeError):
    def __init__(self, fn, status, detail):
        super().__init__(f"{fn} failed: {detail} (status {status})")
        self.status = status


TISE_OK = 0
TISE_ERR_INVALID_ARG = -1
TISE_ERR_HIP = -2
TISE_ERR_NO_DEVICE = -3
TISE_ERR_UNSUPPORTED = -4
TISE_FRECHET_OUT_DOUBLES = 8
TISE_FLAG_NONFINITE = 1
TISE_FLAG_RANK_DEFICIENT = 2

# name -> (restype, argtypes); mirrors include/tise_hip.h one to one
SIGNATURES = {
    "tise_status_string": (c_char_p, [c_int]),
    "tise_last_hip_error": (c_int, []),
    "tise_version": (c_int, []),
    "tise_device_info": (c_int, [POINTER(c_int), POINTER(c_int), POINTER(c_size_t)]),
    "tise_host_register": (c_int, [c_void_p, c_size_t]),
    "tise_host_unregister": (c_int, [c_void_p]),
    "tise_memcpy_h2d_async": (c_int, [c_void_p, c_void_p, c_size_t, c_void_p]),
    "tise_png_unfilter_rgb8": (c_int, [c_void_p, c_int64, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "tise_resize_bilinear_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                         POINTER(c_float), c_void_p, c_void_p]),
    "tise_resize_u8": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_int, c_int, c_int,
                                POINTER(c_float), c_void_p, c_int, c_void_p]),
    "tise_cosine_top1": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_int, c_int, c_float, c_void_p,
                                  c_void_p, c_void_p]),
    "tise_gemm_f16": (c_int, [c_void_p, c_int64, c_void_p, c_int64, c_void_p, c_void_p, c_int64, c_void_p, c_int64, c_int, c_int,
                               c_int, c_int, c_void_p]),
    "tise_layernorm_f16": (c_int, [c_void_p, c_int64, c_void_p, c_void_p, c_void_p, c_int64, c_int64, c_int, c_float, c_void_p]),
    "tise_attention_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tise_patchify_f16": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tise_vit_tokens_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tise_text_tokens_f16": (c_int, [c_void_p, c_void_p, c_void_p, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "tise_gather_rows_f16": (c_int, [c_void_p, c_void_p, c_int64, c_int, c_void_p, c_void_p]),
    "tise_stats_create": (c_int, [c_int, POINTER(c_void_p)]),
    "tise_stats_destroy": (c_int, [c_void_p]),
    "tise_stats_reset": (c_int, [c_void_p, c_void_p]),
    "tise_stats_update": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "tise_stats_update_cov": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "tise_stats_update_sum": (c_int, [c_void_p, c_void_p, c_int64, c_int64, c_void_p]),
    "tise_stats_update_grouped": (c_int, [POINTER(c_void_p), c_int, c_void_p, POINTER(c_int64), c_int64, c_void_p]),
    "tise_stats_buffer": (c_int, [c_void_p, POINTER(c_void_p), POINTER(c_size_t)]),
    "tise_stats_finalize": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p]),
    "tise_frechet_create": (c_int, [c_int, POINTER(c_void_p)]),
    "tise_frechet_destroy": (c_int, [c_void_p]),
    "tise_frechet_distance": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_double, c_void_p, c_void_p]),
    "tise_frechet_prefactor": (c_int, [c_void_p, c_void_p, c_void_p]),
    "tise_frechet_distance_prefactored": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tise_frechet_prefactor_ms": (c_int, [c_void_p, POINTER(c_double)]),
    "tise_frechet_set_profiling": (c_int, [c_void_p, c_int]),
    "tise_frechet_phase_ms": (c_int, [c_void_p, POINTER(c_double), POINTER(c_int)]),
    "tise_eigvalsh": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p]),
    "tise_pivoted_cholesky": (c_int, [c_void_p, c_void_p, c_void_p, POINTER(c_int), c_void_p]),
    "tise_is_update": (c_int, [c_void_p, c_int64, c_int64, c_int, c_double, c_int, c_int64, c_int64, c_int, c_int,
                                c_void_p, c_void_p, c_void_p]),
    "tise_is_finalize": (c_int, [c_void_p, c_int, c_int64, c_int, c_int, c_void_p, c_void_p]),
    "tise_bias_relu_nhwc": (c_int, [c_void_p, c_int64, c_int, c_int64, c_int, c_void_p, c_void_p, c_int64, c_int, c_void_p]),
    "tise_avgpool3_bias_relu_nhwc": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p,
                                              c_int64, c_int, c_void_p]),
    "tise_maxpool3s2_nhwc": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p, c_void_p, c_int64,
                                      c_int, c_void_p]),
    "tise_avgpool3_bias_relu_split_nhwc": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                                    c_void_p, c_int64, c_int, c_void_p]),
    "tise_maxpool3s2_split_nhwc": (c_int, [c_void_p, c_int64, c_int, c_int, c_int, c_int, c_int, c_void_p,
                                            c_int64, c_int, c_void_p]),
    "tise_stem_conv3x3s2_split": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p, c_void_p]),
    "tise_stem_conv3x3s2_split_u8": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                              c_void_p]),
    "tise_stem_conv3x3s2_split_u8_mfma": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p,
                                                   c_void_p, c_void_p]),
    "tise_split_mean_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p]),
    "tise_split_mean_both_nhwc": (c_int, [c_void_p, c_int, c_int, c_int, c_void_p, c_void_p, c_void_p]),
    "tise_conv_split_f16": (c_int, [c_void_p, c_int, c_void_p]),
    "tise_split_overflow_check": (c_int, [POINTER(c_int), c_void_p]),
    "tise_gemm_f64": (c_int, [c_void_p, c_int64, c_int64, c_void_p, c_int64, c_int64, c_void_p, c_int64,
                               c_int, c_int, c_int, c_void_p]),
}

_lib = None


def load():
    """Load the shared library (once).  Raises TiseLibraryError when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise TiseLibraryError(
            f"{LIB_PATH} not found: build it with `python -m tise_toolbox_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:
        raise TiseLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise TiseLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(fn_name, status):
    if status != TISE_OK:
        lib = load()
        detail = lib.tise_status_string(status).decode()
        if status == TISE_ERR_HIP:
            detail += f" (hipError_t {lib.tise_last_hip_error()})"
        raise TiseStatusError(fn_name, status, detail)


def call(fn_name, *args):
    lib = load()
    status = getattr(lib, fn_name)(*args)
    check(fn_name, status)
