"""Frozen outputs of the CPU oracle for the real-width GPU tests.

The parity tests at BASELINE.json's own shapes compare the HIP path with oracle/*.py on seeded synthetic weights.  At those shapes
the ORACLE is what takes the time (286 batch-1-equivalent forwards of a 650M model on 16 host threads: 160 s; Tranception-L at
n_ctx 1024 in fp32 and fp64: 100 s ...), and it computes the same numbers on every run: weights, sequences and mutants are
functions of fixed seeds.  ``cached(name, parts, compute)`` therefore keeps the oracle's outputs under tests/golden/frozen/<name>.npz
together with a fingerprint of everything they depend on (a sha256 over the weight blob -- sampled: both ends and every 4099th
element --, the sequences, positions and dtypes).  A test whose inputs still hash to the stored fingerprint reads the file; any
change to the generator, the seeds or the case recomputes live (and says so).  The files are written by the tests themselves:

    PGMI_FREEZE_DIR=gpurun_out/frozen python -m pytest tests -m gpu        # then: cp gpurun_out/frozen/*.npz tests/golden/frozen/

(the oracle is pinned to the unmodified reference by tests/test_oracle_pinning.py, with or without these files).
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FROZEN_DIR = os.path.join(HERE, "golden", "frozen")


def fingerprint(parts) -> bytes:
    h = hashlib.sha256()
    for p in parts:
        if isinstance(p, np.ndarray):
            a = np.ascontiguousarray(p).reshape(-1)
            h.update(f"{a.dtype}:{a.size}:".encode())
            if a.size <= 1 << 20:
                h.update(a.tobytes())
            else:
                h.update(a[: 1 << 18].tobytes())
                h.update(a[-(1 << 18):].tobytes())
                h.update(np.ascontiguousarray(a[::4099]).tobytes())
        else:
            h.update(repr(p).encode())
        h.update(b"|")
    return h.digest()


def cached(name: str, parts, compute):
    """``compute() -> {key: array}``.  Returns the stored arrays when tests/golden/frozen/<name>.npz carries the fingerprint of
    ``parts``; otherwise calls ``compute`` (and, under PGMI_FREEZE_DIR, writes <name>.npz there)."""
    key = np.frombuffer(fingerprint(parts), dtype=np.uint8)
    path = os.path.join(FROZEN_DIR, name + ".npz")
    if os.path.exists(path) and not os.environ.get("PGMI_FREEZE_DIR"):
        z = np.load(path)
        if np.array_equal(z["fingerprint"], key):
            return {k: z[k] for k in z.files if k != "fingerprint"}
        print(f"[frozen] {name}: the inputs no longer hash to the stored fingerprint -- running the oracle live", file=sys.stderr)
    out = {k: np.asarray(v) for k, v in compute().items()}
    dest = os.environ.get("PGMI_FREEZE_DIR")
    if dest:
        os.makedirs(dest, exist_ok=True)
        np.savez_compressed(os.path.join(dest, name + ".npz"), fingerprint=key, **out)
    return out
