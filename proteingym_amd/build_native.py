"""Builds libpgmi.so (hand-written HIP for gfx950) in-tree with hipcc.

    python -m proteingym_amd.build_native [--force]

hipcc cross-compiles without a GPU; the resulting proteingym_amd/libpgmi.so is git-ignored but
travels to the GPU box with the repo snapshot.
"""
from __future__ import annotations

import concurrent.futures as cf
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libpgmi.so")
BUILD = os.path.join(HERE, "csrc", "_build")
SOURCES = ["api_model.hip", "api_esm.hip", "api_tranception.hip", "api_msa.hip", "api_host.hip", "api_ops.hip", "elementwise.hip", "gemm_f32.hip", "gemm_f16.hip", "attention_f32.hip", "attention_f16.hip", "attention_f16_v3.hip", "attention_f16_prep.hip", "msa_weights.hip", "msa_transformer.hip"]
# attention_f16.hip: keep the MFMA accumulators in ArchVGPRs.  hipcc put the running O / S accumulators into AGPRs and then
# paid 64 v_accvgpr_read + 64 v_accvgpr_write around every online-softmax rescale and around the S -> P conversion (VALU work
# on accumulator data); the kernel fits 184 VGPRs at the same occupancy without them.
EXTRA_FLAGS = {"attention_f16.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"], "attention_f16_v3.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function", "-Wno-unused-value", "-Wno-unused-result"]


def _hipcc() -> str:
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    raise RuntimeError("hipcc not found")


def _digest() -> str:
    h = hashlib.sha256()
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for fn in sorted(os.listdir(root)):
            p = os.path.join(root, fn)
            if os.path.isfile(p) and fn.split(".")[-1] in ("hip", "h", "cpp"):
                h.update(fn.encode())
                h.update(open(p, "rb").read())
    h.update(" ".join(FLAGS).encode())
    h.update(repr(sorted(EXTRA_FLAGS.items())).encode())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(BUILD, exist_ok=True)
    stamp = os.path.join(BUILD, "stamp")
    dig = _digest()

    def fresh():
        return os.path.exists(OUT) and os.path.exists(stamp) and open(stamp).read() == dig
    if not force and fresh():
        return OUT
    # one builder at a time: the ranks of a multi-GPU launch all come through here, and on a box where the library was not
    # built yet they must not write the same object files together -- the others wait for the lock and find the stamp
    import fcntl
    with open(os.path.join(BUILD, "lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        if not force and fresh():
            return OUT
        return _build_locked(dig, stamp, verbose)


def _build_locked(dig: str, stamp: str, verbose: bool) -> str:
    hipcc = _hipcc()

    def compile_one(src):
        obj = os.path.join(BUILD, src.replace(".", "_") + ".o")
        cmd = [hipcc, *FLAGS, *EXTRA_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{r.stdout}\n{r.stderr}")
        if verbose and r.stderr.strip():
            print(r.stderr, file=sys.stderr)
        return obj

    with cf.ThreadPoolExecutor(max_workers=min(8, len(SOURCES))) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"link failed:\n{r.stdout}\n{r.stderr}")
    # a shared object links with unresolved symbols; resolve them now (RTLD_NOW) so a kernel whose host stub the
    # compiler dropped fails the build instead of the first import on the GPU box
    r = subprocess.run([sys.executable, "-c", f"import ctypes, os; ctypes.CDLL({OUT!r}, mode=os.RTLD_NOW)"],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"{OUT} does not load:\n{r.stderr}")
    open(stamp, "w").write(dig)
    if verbose:
        print(f"built {OUT}")
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
