"""List VGPR / SGPR / LDS / spill counts of every kernel of one translation unit.
usage: python tools/kernel_regs.py factorizer_amd/csrc/gemm_bx.hip [substring]"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from factorizer_amd import build as B  # noqa: E402

f = sys.argv[1]
sub = sys.argv[2] if len(sys.argv) > 2 else ""
with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
    cmd = [B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(os.path.basename(f), []), "--cuda-device-only", "-S", f, "-o", tmp.name]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    txt = open(tmp.name).read()
for blk in txt.split("  - .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
    name = subprocess.run(["c++filt", g("name")], capture_output=True, text=True).stdout.strip()
    if sub in name:
        print(f"vgpr {g('vgpr_count'):>4} spill {g('vgpr_spill_count'):>3} sgpr {g('sgpr_count'):>3} lds {g('group_segment_fixed_size'):>6} "
              f"scratch {g('private_segment_fixed_size'):>5}  {name[:110]}")
