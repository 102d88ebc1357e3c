"""CPU: libtise_hip.so loads and exports exactly what include/tise_hip.h declares (no compute calls)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "tise_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(tise_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    from tise_toolbox_amd import _lib
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_lib.SIGNATURES) == declared, "tise_toolbox_amd/_lib.py SIGNATURES must mirror include/tise_hip.h"


def test_library_loads_and_exports_every_symbol():
    from tise_toolbox_amd import _lib, build
    build.build(force=False, verbose=False)            # hipcc cross-compiles gfx950 without a GPU
    lib = _lib.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), name
    assert lib.tise_version() == 1
    assert lib.tise_status_string(0) == b"ok"
    assert lib.tise_status_string(-1) == b"invalid argument"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for name in _declared_symbols():
        getattr(raw, name)


def test_argument_validation_without_a_gpu():
    """Entry points reject bad arguments before touching the device."""
    from tise_toolbox_amd import _lib
    lib = _lib.load()
    h = ctypes.c_void_p()
    assert lib.tise_stats_create(0, ctypes.byref(h)) == _lib.TISE_ERR_INVALID_ARG
    assert lib.tise_frechet_create(-3, ctypes.byref(h)) == _lib.TISE_ERR_INVALID_ARG
    assert lib.tise_stats_update(None, None, 4, 4, None) == _lib.TISE_ERR_INVALID_ARG
    assert lib.tise_is_finalize(None, 10, 10, 10, 0, None, None) == _lib.TISE_ERR_INVALID_ARG
    assert lib.tise_is_update(None, 4, 8, 8, 0.0, 0, 0, 4, 10, 0, None, None, None) == _lib.TISE_ERR_INVALID_ARG
    lut = (ctypes.c_float * 768)()
    assert lib.tise_resize_bilinear_u8(None, 1, 0, 5, None, 299, 299, 1, lut, None, None) == _lib.TISE_ERR_INVALID_ARG


def test_product_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import numpy as np
    from tise_toolbox_amd import _lib, fid_score, inception_score
    with pytest.raises(_lib.TiseLibraryError):
        fid_score.calculate_frechet_distance(np.zeros(2), np.eye(2), np.zeros(2), np.eye(2))
    with pytest.raises(_lib.TiseLibraryError):
        inception_score.inception_score_from_logits(np.zeros((4, 3), np.float32))
    with pytest.raises(_lib.TiseLibraryError):
        fid_score.get_activations([], None, cuda=False)


def test_product_never_imports_the_oracle():
    """The product package must not route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "tise_toolbox_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), f
                assert "from tests" not in text, f


def test_png_library_exports_its_header():
    """libtise_png.so (gcc, csrc/png_decode.c): every symbol include/tise_png.h declares, bound as _png_worker binds them."""
    from tise_toolbox_amd import _png_worker, build
    build.build_png(force=False, verbose=False)
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tise_png.h")).read(), flags=re.S)
    declared = sorted(set(re.findall(r"\b(tise_png_[a-z0-9_]+)\s*\(", text)))
    assert declared == ["tise_png_decode_rgb8", "tise_png_inflate_backend", "tise_png_inflate_slot", "tise_png_probe",
                        "tise_png_scratch_bytes", "tise_png_slot_bytes"]
    raw = ctypes.CDLL(build.PNG_LIB)
    for name in declared:
        getattr(raw, name)
    lib = _png_worker.load_decoder()
    assert lib is not None and lib.tise_png_inflate_backend() in (0, 1)
