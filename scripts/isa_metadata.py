"""Register / LDS / spill metadata of every kernel in libpgmi.so's sources, from hipcc's own assembly (gfx950 code-object notes).

    python scripts/isa_metadata.py > profiles/r3/isa_metadata.txt        # no GPU needed (hipcc cross-compiles)

Columns: arch VGPRs, AGPRs, SGPRs, SGPR / VGPR spills, scratch bytes per lane, static LDS bytes, MFMA instructions in the body."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proteingym_amd import build_native as bn  # noqa: E402


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def main():
    print(f"build digest {bn._digest()[:16]}  flags: {' '.join(bn.FLAGS)}")
    with tempfile.TemporaryDirectory() as d:
        for src in bn.SOURCES:
            asm = os.path.join(d, src + ".s")
            subprocess.run([bn._hipcc(), *bn.FLAGS, *bn.EXTRA_FLAGS.get(src, []), "-S", "--cuda-device-only", "-w",
                            os.path.join(bn.CSRC, src), "-o", asm], check=True, capture_output=True)
            text = open(asm).read()
            mfma = {}
            for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*s_endpgm", text, re.S | re.M):
                mfma[m.group(1)] = len(re.findall(r"^\s*v_mfma", m.group(2), re.M))
            kernels = re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s*\d+", text, re.S)
            rows = []
            for k in kernels:
                g = lambda key: (re.search(rf"\.{key}:\s*(\S+)", k) or [None, "?"])[1]  # noqa: E731
                rows.append((g("name"), g("vgpr_count"), g("agpr_count"), g("sgpr_count"), g("sgpr_spill_count"), g("vgpr_spill_count"),
                             g("private_segment_fixed_size"), g("group_segment_fixed_size")))
            names = demangle([r[0] for r in rows])
            print(f"\n== {src}")
            print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'sSpill':>6} {'vSpill':>6} {'scratch':>7} {'LDS':>7} {'MFMA':>5}  kernel")
            for r in rows:
                nm = re.sub(r"\(.*", "", names.get(r[0], r[0]))
                print(f"{r[1]:>5} {r[2]:>5} {r[3]:>5} {r[4]:>6} {r[5]:>6} {r[6]:>7} {r[7]:>7} {mfma.get(r[0], 0):>5}  {nm}")


if __name__ == "__main__":
    main()
