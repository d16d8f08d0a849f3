"""ctypes binding of libqinco_hip.so (include/qinco_hip.h).  No CPU fallback: a missing library raises."""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "libqinco_hip.so"

X_F32, X_U8 = 0, 1
CODE_I64, CODE_I32, CODE_U8 = 0, 1, 2
FLAG_NORMALISED = 1
CREATE_SPLIT_F16, CREATE_IVF_FP32, CREATE_TABLE_VALU, CREATE_DECODE_FOLDED, CREATE_TABLE_NO_COOP = 1, 2, 4, 8, 16
CREATE_SPLIT_NO_CALIBRATION, CREATE_NO_PRESEL_FUSION, CREATE_NO_SMALL_LAUNCH, CREATE_EPILOGUE_SELECT = 32, 64, 128, 256
CREATE_NO_EPILOGUE_SELECT = 512

# every symbol include/qinco_hip.h declares (tests check the library exports all of them)
API_SYMBOLS = [
    "qinco_create", "qinco_create_ex", "qinco_create_opt", "qinco_describe", "qinco_padded_shape", "qinco_load_instance", "qinco_split_stats", "qinco_gather_codes", "qinco_rccl_library", "qinco_destroy", "qinco_set_beam", "qinco_encode", "qinco_decode", "qinco_encode_host",
    "qinco_decode_host", "qinco_profile_enable", "qinco_profile_read", "qinco_profile_read2", "qinco_flops_per_vector_encode",
    "qinco_flops_per_vector_decode", "qinco_shape_supported", "qinco_last_error", "qinco_version",
    "qinco_lut_create", "qinco_lut_destroy", "qinco_lut_decode", "qinco_lut_decode_host",
    "qinco_ivf_last_stats", "qinco_check", "qinco_selftest", "qinco_knn_create", "qinco_knn_destroy", "qinco_knn_search", "qinco_knn_search_host", "qinco_knn_set_option", "qinco_knn_last_stats", "qinco_knn_roles_stats", "qinco_sqerr_sum", "qinco_rerank",
]


class QincoDesc(C.Structure):
    _fields_ = [("D", C.c_int32), ("De", C.c_int32), ("Dh", C.c_int32), ("L", C.c_int32), ("M", C.c_int32),
                ("K", C.c_int32), ("A", C.c_int32), ("B", C.c_int32), ("qinco1_mode", C.c_int32),
                ("ivf_K", C.c_int32), ("max_batch", C.c_int64)]


class QincoOptions(C.Structure):
    _fields_ = [("struct_bytes", C.c_int32), ("create_flags", C.c_int32), ("mlp_P", C.c_int32), ("mlp_var", C.c_int32),
                ("table_coop_max", C.c_int64)]


class QincoSplitReport(C.Structure):
    _fields_ = [("split_form", C.c_int32), ("calibrated", C.c_int32), ("calib_vectors", C.c_int32),
                ("calib_rows_differing", C.c_int32), ("calib_max_rel_err", C.c_float), ("overflowed", C.c_int32),
                ("lo_sampled", C.c_int64), ("lo_subnormal", C.c_int64)]


FP = C.POINTER(C.c_float)
FPP = C.POINTER(FP)


class QincoWeights(C.Structure):
    _fields_ = [("data_mean", FP), ("data_std", C.c_float), ("codebook", FPP), ("sub_codebook", FPP),
                ("in_proj", FPP), ("out_proj", FPP), ("cat_w", FPP), ("cat_b", FPP), ("up", FPP), ("down", FPP)]


class QincoLibraryError(RuntimeError):
    pass


_lib = None


def _preload_hip_runtime() -> None:
    """Guarantee ONE HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME
    libamdhip64.so.7, same as /opt/rocm's).  If libqinco_hip.so pulled in /opt/rocm's copy and torch later
    loaded its own (or the other way round), the second runtime finds no GPU ("No HIP GPUs are available").
    Loading torch's copy by its exact path first makes both libqinco_hip.so (matched by SONAME) and a later
    `import torch` (matched by inode) share it; without torch installed /opt/rocm's runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already loaded its runtime; ours resolves to it by SONAME
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = Path(list(spec.submodule_search_locations)[0]) / "lib" / "libamdhip64.so"
    if cand.exists():
        try:
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load() -> C.CDLL:
    """Load the HIP library.  Fails loudly (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("QINCO_HIP_LIB", LIB_PATH))
    if not path.exists():
        raise QincoLibraryError(
            f"{path} not found: build it with `python -m qinco_amd.build` (hipcc, --offload-arch=gfx950). "
            "qinco_amd has no CPU fallback.")
    _preload_hip_runtime()
    try:
        lib = C.CDLL(str(path))
    except OSError as e:  # e.g. libamdhip64 missing
        raise QincoLibraryError(f"cannot load {path}: {e}") from e
    vp, i32, i64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_double
    lib.qinco_create.argtypes = [C.POINTER(QincoDesc), C.POINTER(QincoWeights), C.POINTER(vp)]
    lib.qinco_create_ex.argtypes = [C.POINTER(QincoDesc), C.POINTER(QincoWeights), C.c_int32, C.POINTER(vp)]
    lib.qinco_create_opt.argtypes = [C.POINTER(QincoDesc), C.POINTER(QincoWeights), C.POINTER(QincoOptions), C.POINTER(vp)]
    lib.qinco_create_opt.restype = C.c_int
    lib.qinco_padded_shape.argtypes = [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    lib.qinco_padded_shape.restype = C.c_int
    lib.qinco_load_instance.argtypes = [C.c_char_p]
    lib.qinco_load_instance.restype = C.c_int
    lib.qinco_gather_codes.argtypes = [vp, i64, C.c_int32, i32, vp, C.POINTER(i64), C.c_int32, C.c_int32, C.c_int32, vp, vp]
    lib.qinco_gather_codes.restype = C.c_int
    lib.qinco_rccl_library.argtypes = [C.c_char_p, C.c_size_t]
    lib.qinco_rccl_library.restype = C.c_int
    lib.qinco_split_stats.argtypes = [vp, C.POINTER(QincoSplitReport)]
    lib.qinco_split_stats.restype = C.c_int
    lib.qinco_describe.argtypes = [vp, C.c_char_p, C.c_int32]
    lib.qinco_describe.restype = C.c_int
    lib.qinco_destroy.argtypes = [vp]
    lib.qinco_set_beam.argtypes = [vp, C.c_int32, C.c_int32]
    lib.qinco_encode.argtypes = [vp, vp, i32, i64, i64, vp, i32, vp, i32, vp]
    lib.qinco_decode.argtypes = [vp, vp, i32, i64, vp, i32, vp]
    lib.qinco_encode_host.argtypes = [vp, vp, i32, i64, i64, vp, i32, vp, i32]
    lib.qinco_decode_host.argtypes = [vp, vp, i32, i64, vp, i32]
    lib.qinco_check.argtypes = [vp, vp]
    lib.qinco_selftest.argtypes = []
    lib.qinco_selftest.restype = C.c_int
    lib.qinco_check.restype = C.c_int
    lib.qinco_profile_enable.argtypes = [vp, i32]
    lib.qinco_profile_read.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl)]
    lib.qinco_profile_read2.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl), C.POINTER(dbl)]
    lib.qinco_profile_read2.restype = C.c_int
    lib.qinco_flops_per_vector_encode.argtypes = [vp]
    lib.qinco_flops_per_vector_encode.restype = dbl
    lib.qinco_flops_per_vector_decode.argtypes = [vp]
    lib.qinco_flops_per_vector_decode.restype = dbl
    lib.qinco_shape_supported.argtypes = [C.c_int32, C.c_int32, C.c_int32]
    lib.qinco_ivf_last_stats.argtypes = [vp, C.POINTER(i64), C.POINTER(C.c_int32)]
    I32P = C.POINTER(C.c_int32)
    lib.qinco_lut_create.argtypes = [FP, C.c_int32, i64, C.c_int32, I32P, I32P, i64, C.POINTER(vp)]
    lib.qinco_lut_destroy.argtypes = [vp]
    lib.qinco_lut_decode.argtypes = [vp, vp, i32, C.c_int32, i64, vp, vp]
    lib.qinco_lut_decode_host.argtypes = [vp, vp, i32, C.c_int32, i64, vp]
    lib.qinco_knn_create.argtypes = [C.c_int32, C.POINTER(vp)]
    lib.qinco_knn_destroy.argtypes = [vp]
    lib.qinco_knn_search.argtypes = [vp, vp, i64, vp, i64, C.c_int32, vp, vp, vp]
    lib.qinco_knn_search_host.argtypes = [vp, vp, i64, vp, i64, C.c_int32, vp, vp]
    lib.qinco_knn_set_option.argtypes = [vp, C.c_int32, i64]
    lib.qinco_knn_roles_stats.argtypes = [vp, C.POINTER(i64)]
    lib.qinco_knn_roles_stats.restype = C.c_int
    lib.qinco_knn_last_stats.argtypes = [vp, C.POINTER(i64)]
    lib.qinco_sqerr_sum.argtypes = [vp, vp, i64, C.POINTER(dbl), vp]
    lib.qinco_rerank.argtypes = [vp, vp, i64, C.c_int32, C.c_int32, C.c_int32, vp, vp, C.c_int32, vp, vp, vp, vp, vp]
    lib.qinco_rerank.restype = C.c_int
    for name in ("qinco_lut_create", "qinco_lut_destroy", "qinco_lut_decode", "qinco_lut_decode_host"):
        getattr(lib, name).restype = C.c_int
    lib.qinco_last_error.restype = C.c_char_p
    lib.qinco_version.restype = C.c_char_p
    for name in ("qinco_create", "qinco_destroy", "qinco_set_beam", "qinco_encode", "qinco_decode",
                 "qinco_encode_host", "qinco_decode_host", "qinco_profile_enable", "qinco_profile_read",
                 "qinco_shape_supported"):
        getattr(lib, name).restype = C.c_int
    _lib = lib
    return lib


def check(rc: int) -> None:
    """Map status codes to the exceptions the reference raises at the same sites."""
    if rc == 0:
        return
    msg = (load().qinco_last_error() or b"").decode()
    if rc == -1:
        raise ValueError(msg)          # reference: assert / ValueError (utils.py:169-172, qinco_base.py:525-526)
    if rc == -3:
        raise NotImplementedError(msg)
    if rc == -4:
        raise IndexError(msg)          # torch indexing error on out-of-range codes
    raise RuntimeError(msg)
