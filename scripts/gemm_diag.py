"""Phase timing of the persistent ping-pong GEMM: builds and runs tools/gemm_diag.hip (the tuning-only instantiations of
proteingym_amd/csrc/gemm16x_kernel.h live there, not in libpgmi.so) and prints, for the early and late wave of SIMD 0 of
workgroup 0, the mean shader clocks spent working in / waiting at the end of each of the four phases.

    python scripts/gemm_diag.py [flags ...]          (default: 1000 = the shipped DMA form; see tools/gemm_diag.hip)
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exe = os.path.join(ROOT, "tools", "gemm_diag")
src = os.path.join(ROOT, "tools", "gemm_diag.hip")
hdr = os.path.join(ROOT, "proteingym_amd", "csrc", "gemm16x_kernel.h")
if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(src), os.path.getmtime(hdr)):
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I", os.path.join(ROOT, "include"), "-o", exe, src])
for flags in (sys.argv[1:] or ["1000"]):
    subprocess.check_call([exe, flags])
