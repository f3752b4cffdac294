"""Count, per translation unit of the BUILT library, the packed-fp32 VALU instructions whose LOW result takes a HIGH source
half (`v_pk_{add,mul,fma}_f32 ... op_sel:[..1..]`).  Four such instructions — formed by the SLP vectorizer in a store
epilogue — are what made the eight-wave upcat kernel stop replaying bit for bit beside MFMA waves on gfx950
(assembly-level bisect: profiles/r04_nondeterminism.md).  The build therefore compiles with -fno-slp-vectorize
(factorizer_amd/build.py) and tests/test_no_spills.py asserts that this audit finds none, in any unit.
Reads the objects in factorizer_amd/csrc/build (llvm-objdump of the embedded gfx950 code object; no recompilation).
usage: python tools/pk_opsel_audit.py"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "factorizer_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
PAT = re.compile(r"\bv_pk_(add|mul|fma)_f32\b.*\bop_sel:\[[01,]*1[01,]*\]")


def count(obj):
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "co.elf")
        if subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True).returncode:
            return 0, 0   # no device code
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    return sum(1 for ln in txt.splitlines() if PAT.search(ln)), txt.count("v_pk_")


def audit():
    return {f[:-2]: count(os.path.join(OBJ, f)) for f in sorted(os.listdir(OBJ)) if f.endswith(".o")}


if __name__ == "__main__":
    for tu, (n, npk) in audit().items():
        print(f"{tu:16s} {n:6d} op_sel'd packed-fp32 instructions ({npk} v_pk_* in all)")
