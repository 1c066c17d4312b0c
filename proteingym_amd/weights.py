"""Host-side mirror of ``proteingym/utils/weights.py`` (sequence weights of an alignment) backed by
the HIP pair-count kernel in libpgmi.so (``pgmi_msa_cluster_counts``, csrc/msa_weights.hip).

Same function names and arguments as the reference (weights.py:13-53 ``calc_weights_fast``, :56-61
``is_empty_sequence_matrix``, :64-93 ``map_from_alphabet``, :96-111 ``map_matrix``); ``num_cpus`` is
accepted and ignored, ``device`` is additive.  There is no CPU path here: without the library or a
GPU the call raises ``PgmiError``.
"""
from __future__ import annotations

import ctypes as C
from collections import defaultdict

import numpy as np

from . import _lib
from ._lib import PgmiError


def is_empty_sequence_matrix(matrix, empty_value):
    assert len(matrix.shape) == 2, f"Matrix must be 2D; shape={matrix.shape}"
    assert isinstance(empty_value, (int, float)), f"empty_value must be a number; type={type(empty_value)}"
    return np.all((matrix == empty_value), axis=1)


def map_from_alphabet(alphabet, default):
    map_ = {c: i for i, c in enumerate(alphabet)}
    try:
        default = map_[default]
    except KeyError:
        raise ValueError("Default {} is not in alphabet {}".format(default, alphabet))
    return defaultdict(lambda: default, map_)


def map_matrix(matrix, map_):
    return np.vectorize(map_.__getitem__)(matrix)


def num_cluster_members(matrix_mapped, identity_threshold, invalid_value, device=0, return_ms=False):
    """calc_num_cluster_members_nogaps_parallel (weights.py:164-216) for ALL rows of the matrix:
    int32 counts, self included; 0 for rows without a valid symbol."""
    m = np.asarray(matrix_mapped)
    if m.ndim != 2:
        raise ValueError(f"Matrix must be 2D; shape={m.shape}")
    if m.size and (m.min() < -128 or m.max() > 127):
        raise ValueError("mapped symbols must fit int8")
    m8 = np.ascontiguousarray(m, dtype=np.int8)
    n, l = m8.shape
    out = np.zeros(n, dtype=np.int32)
    ms = C.c_double(0.0)
    lib = _lib.load()
    rc = lib.pgmi_msa_cluster_counts(int(device), m8.ctypes.data_as(C.POINTER(C.c_int8)), n, l, int(invalid_value),
                                     float(identity_threshold), out.ctypes.data_as(C.POINTER(C.c_int32)), C.byref(ms))
    _lib.check(rc)
    return (out, ms.value) if return_ms else out


def calc_weights_fast(matrix_mapped, identity_threshold, empty_value, num_cpus=1, device=0):
    """weights.py:13-53: weight = 1 / cluster size, 0 for empty sequences."""
    empty_idx = is_empty_sequence_matrix(matrix_mapped, empty_value=empty_value)
    counts = num_cluster_members(matrix_mapped, identity_threshold, empty_value, device=device)
    weights = np.zeros(matrix_mapped.shape[0])
    weights[~empty_idx] = 1.0 / counts[~empty_idx]
    return weights
