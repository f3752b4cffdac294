"""Which kernels of the built library use scratch (a private segment: spilled registers or dynamically indexed arrays)?
Reads the kernel metadata notes of the objects in factorizer_amd/csrc/build (no recompilation).
usage: python tools/scratch_audit.py [-v]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "factorizer_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernels_of(obj):
    """[(kernel name, private_segment_fixed_size, vgpr_spill_count, vgpr_count)] of one host object with an embedded gfx950 code object"""
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "co.elf")
        r = subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], capture_output=True)
        if r.returncode != 0:   # a translation unit without device code (api.hip)
            return []
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        txt = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], check=True, capture_output=True, text=True).stdout
    out = []
    for blk in txt.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "0"])[1]  # noqa: E731
        out.append((g("name"), int(g("private_segment_fixed_size")), int(g("vgpr_spill_count")), int(g("vgpr_count"))))
    return out


def audit():
    res = {}
    for f in sorted(os.listdir(OBJ)):
        if f.endswith(".o"):
            res[f[:-2]] = kernels_of(os.path.join(OBJ, f))
    return res


if __name__ == "__main__":
    for tu, ks in audit().items():
        bad = [k for k in ks if k[1] > 0]
        print(f"{tu:16s} {len(ks):4d} kernels, {len(bad):4d} with scratch")
        if "-v" in sys.argv:
            for n, ps, sp, vg in bad:
                dem = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
                print(f"      scratch {ps:5d} B, {sp:4d} spilled VGPRs: {dem[:120]}")
