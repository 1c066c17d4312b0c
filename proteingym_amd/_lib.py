"""ctypes binding of libpgmi.so (include/pgmi.h).  No torch import here: the GPU path is
Python -> ctypes -> C ABI -> HIP kernels.  There is no CPU fallback: if the library is missing
or no GPU is visible the calls raise."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libpgmi.so")

ABI_VERSION = 4
ARCH_ESM1B, ARCH_ESM2, ARCH_TRANCEPTION = 1, 2, 3
PREC_FP32, PREC_BF16, PREC_F16X3 = 0, 1, 2
PRECISIONS = {"fp32": PREC_FP32, "bf16": PREC_BF16, "f16x3": PREC_F16X3}
K_NAMES = ["embed", "layernorm", "gemm_qkv", "attention", "gemm_out", "gemm_fc1", "gemm_fc2",
           "head", "score", "kept_rows"]


EINVAL, ENOMEM, EHIP, ENODEV, EPARSE, EOVERFLOW = -1, -2, -3, -4, -5, -6          # include/pgmi.h PGMI_E*


class PgmiError(RuntimeError):
    """A failing C call; ``code`` is its PGMI_E* return value (None when the failure is the binding's own)."""

    def __init__(self, message, code=None):
        super().__init__(message)
        self.code = code


class Config(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "abi_version", "arch", "layers", "embed_dim", "heads", "ffn_dim", "vocab",
        "max_positions", "token_dropout", "emb_layer_norm_before", "precision", "max_rows")] + [("ln_eps", C.c_float)]


_lib = None

_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)
_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)

# (name, restype, argtypes) -- must list every symbol include/pgmi.h declares
SIGNATURES = [
    ("pgmi_abi_version", C.c_int, []),
    ("pgmi_device_count", C.c_int, []),
    ("pgmi_last_error", C.c_char_p, []),
    ("pgmi_weight_count", C.c_int64, [C.POINTER(Config)]),
    ("pgmi_model_create", C.c_int, [C.POINTER(Config), _f32p, C.c_int64, C.c_int, C.POINTER(C.c_void_p)]),
    ("pgmi_model_destroy", None, [C.c_void_p]),
    ("pgmi_model_device", C.c_int, [C.c_void_p]),
    ("pgmi_token_logprobs", C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int, _f32p]),
    ("pgmi_masked_logprobs", C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, C.c_int, _f32p]),
    ("pgmi_assay_create", C.c_int, [C.c_void_p, _i32p, C.c_int, _i32p, C.c_int, C.c_int,
                                    _i32p, _i32p, _i32p, _i64p, C.c_int64, C.POINTER(C.c_void_p)]),
    ("pgmi_assay_run", C.c_int, [C.c_void_p, C.c_void_p, _f64p, _f32p, C.c_void_p]),
    ("pgmi_assay_destroy", None, [C.c_void_p]),
    ("pgmi_pppl_create", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), _i64p, C.c_int64, C.POINTER(C.c_void_p)]),
    ("pgmi_pppl_run", C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, _f64p, _f32p, C.c_void_p]),
    ("pgmi_pppl_rows", C.c_int64, [C.c_void_p, C.c_int64, C.c_int64]),
    ("pgmi_pppl_stats", C.c_int, [C.c_void_p, _i64p, _i64p, _i64p, _i64p]),
    ("pgmi_pppl_destroy", None, [C.c_void_p]),
    ("pgmi_parse_mutants", C.c_int, [C.c_char_p, _i64p, C.c_int64, C.c_char_p, C.c_int, C.c_int,
                                     _i32p, _i32p, _i32p, _i64p, _i64p]),
    ("pgmi_score_mutants", C.c_int, [_f32p, C.c_int, C.c_int, _i32p, _i32p, _i32p, _i64p, C.c_int64, _f64p]),
    ("pgmi_optimal_window", None, [C.c_int, C.c_int, C.c_int, _i32p, _i32p]),
    ("pgmi_profile_enable", C.c_int, [C.c_void_p, C.c_int]),
    ("pgmi_profile_get", C.c_int, [C.c_void_p, C.c_int, _f64p, _i64p, _f64p, _f64p]),
    ("pgmi_profile_reset", C.c_int, [C.c_void_p]),
    ("pgmi_synchronize", C.c_int, [C.c_void_p]),
    ("pgmi_set_option", C.c_int, [C.c_char_p, C.c_int64]),
    ("pgmi_op_layernorm", C.c_int, [C.c_int, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_float, _f32p]),
    ("pgmi_op_gemm", C.c_int, [C.c_int, C.c_int, _f32p, _f32p, _f32p, _f32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    ("pgmi_tr_token_logprobs", C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int, _f32p]),
    ("pgmi_tr_sequence_loglik", C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, C.c_int, _f32p, C.c_int,
                                          _i32p, _i32p, _i32p, _i32p, C.c_float, _f32p]),
    ("pgmi_tr_sequence_loglik_shared", C.c_int, [C.c_void_p, _i32p, _i32p, C.c_int, C.c_int, _f32p, C.c_int,
                                                 _i32p, _i32p, _i32p, _i32p, C.c_float, _f32p, _f32p, _i64p]),
    ("pgmi_bench_gemm", C.c_int, [C.c_int] * 9 + [_f64p]),
    ("pgmi_bench_gemm_ab", C.c_int, [C.c_int] * 7 + [_i32p, C.c_int, C.c_int, C.c_int, _f64p]),
    ("pgmi_op_attention", C.c_int, [C.c_int, C.c_int, _f32p, _i32p, C.c_int, C.c_int, C.c_int, C.c_int, _f32p]),
    ("pgmi_msa_token_logprobs", C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int, _f32p]),
    ("pgmi_msa_masked_logprobs", C.c_int, [C.c_void_p, _i32p, C.c_int, C.c_int, C.c_int, _i32p, _i32p, C.c_int, _f32p]),
    ("pgmi_msa_cluster_counts", C.c_int, [C.c_int, C.POINTER(C.c_int8), C.c_int64, C.c_int64, C.c_int, C.c_double, _i32p, _f64p]),
]


def load():
    """Load libpgmi.so (building it is __graft_entry__.build()'s / build_native's job)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PgmiError(f"{LIB_PATH} not found: run `python -m proteingym_amd.build_native` "
                        "(hipcc, gfx950).  There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SIGNATURES:
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pgmi_abi_version() != ABI_VERSION:
        raise PgmiError("libpgmi ABI version mismatch; rebuild")
    _lib = lib
    return lib


def check(rc: int):
    if rc != 0:
        raise PgmiError(f"libpgmi error {rc}: {load().pgmi_last_error().decode(errors='replace')}", code=int(rc))


def ptr(a: np.ndarray, ty):
    return a.ctypes.data_as(ty) if a is not None else None


def as_i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


def as_f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)
