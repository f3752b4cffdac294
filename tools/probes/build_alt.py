"""Build a second copy of the library with extra per-file flags, for same-box A/B runs:
  python tools/probes/build_alt.py <out.so> <flag> file.hip [file.hip ...]
  FZ_LIB_PATH=<out.so> python bench.py ...
Objects of the listed files are compiled with the extra flag into a scratch directory; every other object is the default build's."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from factorizer_amd import build as B  # noqa: E402

out, flag, files = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]   # several flags: comma-separated
B.build(verbose=False)
alt = os.path.join(B.OBJ, "alt")
os.makedirs(alt, exist_ok=True)
import concurrent.futures as cf  # noqa: E402


def one(src):
    if src not in files:
        return os.path.join(B.OBJ, src[:-4] + ".o")
    obj = os.path.join(alt, src[:-4] + ".o")
    cmd = [B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(src, []), *flag, "-c", os.path.join(B.CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stderr[-3000:])
    return obj


with cf.ThreadPoolExecutor(8) as ex:
    objs = list(ex.map(one, B.sources()))
r = subprocess.run([B._hipcc(), "--offload-arch=" + B.ARCH, "-shared", "-fPIC", "-o", out, *objs], capture_output=True, text=True)
if r.returncode:
    sys.exit(r.stderr[-3000:])
print("built", out)
