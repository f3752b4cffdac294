"""Report gfx950 kernels that spill VGPRs (scratch traffic shows up as extra WRITE_SIZE / FETCH_SIZE and as time:
the fused MLP backward lost 10 % to 32 spilled registers after an innocent-looking template change).
usage: python tools/check_spills.py [file.hip ...]      (default: every translation unit of factorizer_amd/csrc)"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "factorizer_amd", "csrc")
sys.path.insert(0, ROOT)
from factorizer_amd import build as B  # noqa: E402


def main():
    files = sys.argv[1:] or [os.path.join(CSRC, f) for f in B.sources()]
    bad = 0
    for f in files:
        with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
            cmd = [B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(os.path.basename(f), []), "-I", os.path.join(ROOT, "include"),
                   "--cuda-device-only", "-S", f, "-o", tmp.name]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode:
                print(f"{f}: hipcc failed\n{r.stderr[-2000:]}")
                bad += 1
                continue
            txt = open(tmp.name).read()
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
            name, vg, sp = m.group(1), int(m.group(2)), int(m.group(3))
            if sp:
                dem = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
                print(f"{os.path.basename(f)}: {sp:4d} spilled VGPRs ({vg} used)  {dem[:150]}")
                bad += 1
    print(f"{bad} kernel(s) with spills" if bad else "no spills")


if __name__ == "__main__":
    main()
