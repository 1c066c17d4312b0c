"""TEST INFRASTRUCTURE ONLY -- loader for the C restatement in msa_weights_oracle.c plus a small numpy
twin (used to cross-check the C build).  Never imported by proteingym_amd/.

Reference: proteingym/utils/weights.py:13-53 (calc_weights_fast), :164-216 (pair count)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "msa_weights_oracle.c")
OUT_DIR = os.path.join(HERE, "_build")
OUT = os.path.join(OUT_DIR, "libmsa_weights_oracle.so")


def build(force=False):
    if not force and os.path.exists(OUT) and os.path.getmtime(OUT) >= os.path.getmtime(SRC):
        return OUT
    os.makedirs(OUT_DIR, exist_ok=True)
    subprocess.run(["gcc", "-O3", "-fopenmp", "-shared", "-fPIC", "-o", OUT, SRC], check=True)
    return OUT


def cluster_counts(matrix, identity_threshold, invalid_value, threads=None):
    lib = C.CDLL(build())
    m = np.ascontiguousarray(matrix, dtype=np.int8)
    n, l = m.shape
    out = np.zeros(n, dtype=np.int32)
    if threads:
        os.environ["OMP_NUM_THREADS"] = str(threads)
    lib.msa_cluster_counts_oracle.argtypes = [C.POINTER(C.c_int8), C.c_int64, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_int32)]
    lib.msa_cluster_counts_oracle(m.ctypes.data_as(C.POINTER(C.c_int8)), n, l, int(invalid_value), float(identity_threshold),
                                  out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def cluster_counts_numpy(matrix, identity_threshold, invalid_value):
    m = np.asarray(matrix)
    valid = m != invalid_value
    nongap = valid.sum(1)
    out = np.zeros(len(m), dtype=np.int32)
    for i in range(len(m)):
        if nongap[i] == 0:
            continue
        matches = ((m == m[i]) & valid[i]).sum(1)
        keep = matches / nongap[i] > identity_threshold
        keep[i] = True
        out[i] = keep.sum()
    return out


def calc_weights(matrix, identity_threshold, empty_value):
    """weights.py:13-53."""
    counts = cluster_counts(matrix, identity_threshold, empty_value)
    w = np.zeros(len(counts))
    w[counts > 0] = 1.0 / counts[counts > 0]
    return w
