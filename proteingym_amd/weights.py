"""Sequence weights of an alignment (the role of ``proteingym/utils/weights.py``) on the HIP pair-count kernel in
libpgmi.so (``pgmi_msa_cluster_counts``, csrc/msa_weights.hip).

``calc_weights_fast`` keeps the reference's name and arguments (weights.py:13-53; ``num_cpus`` is accepted and ignored,
``device`` is additive).  The reference's EVcouplings helpers that turn letters into symbol indices (a defaultdict applied
through ``np.vectorize``, weights.py:64-111) are replaced by one 256-entry byte table applied to the whole alignment at once
(``encode_alignment``).  There is no CPU path here: without the library or a GPU the call raises ``PgmiError``.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib


def symbol_table(alphabet: str, default: str) -> np.ndarray:
    """uint8[256]: byte of a letter -> its index in ``alphabet``; every other byte -> the index of ``default``."""
    pos = alphabet.find(default)
    if pos < 0 or len(default) != 1:
        raise ValueError(f"Default {default} is not in alphabet {alphabet}")
    table = np.full(256, pos, dtype=np.uint8)
    table[np.frombuffer(alphabet.encode("ascii"), dtype=np.uint8)] = np.arange(len(alphabet), dtype=np.uint8)
    return table


def encode_alignment(sequences, alphabet: str, default: str) -> np.ndarray:
    """int8 [n_sequences, length] symbol indices of equal-length sequences (letters outside the alphabet count as ``default``)."""
    sequences = list(sequences)
    if not sequences:
        return np.zeros((0, 0), dtype=np.int8)
    width = len(sequences[0])
    if any(len(s) != width for s in sequences):
        raise ValueError("alignment rows differ in length")
    raw = np.frombuffer("".join(sequences).encode("latin-1"), dtype=np.uint8).reshape(len(sequences), width)
    return symbol_table(alphabet, default)[raw].astype(np.int8)


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
    matrix_mapped = np.asarray(matrix_mapped)
    if matrix_mapped.ndim != 2:
        raise ValueError(f"Matrix must be 2D; shape={matrix_mapped.shape}")
    counts = num_cluster_members(matrix_mapped, identity_threshold, empty_value, device=device)
    occupied = (matrix_mapped != empty_value).any(axis=1)           # a sequence of nothing but the empty symbol weighs 0
    weights = np.zeros(matrix_mapped.shape[0])
    weights[occupied] = 1.0 / counts[occupied]
    return weights
