"""Build libfactorizer_hip.so (gfx950) in-tree with hipcc.  `python -m factorizer_amd.build`.

The .so is git-ignored but travels to the GPU box with the gpurun snapshot.  hipcc
cross-compiles for gfx950 without a GPU present.
"""
from __future__ import annotations

import concurrent.futures as cf
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "csrc", "build")
LIB = os.path.join(HERE, "libfactorizer_hip.so")
ARCH = "gfx950"
FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-comment",
         "-ffp-contract=fast"]
# -fno-slp-vectorize is the DEFAULT (round 4).  Three measured reasons:
#  * correctness: the SLP vectorizer packs pairs of scalar fp32 adds into `v_pk_add_f32 ... op_sel:[0,1]` (low result from a
#    HIGH source half).  Four of those in a store epilogue made an eight-wave MFMA kernel return run-to-run different values in
#    lanes 48-63 on gfx950; replacing exactly those four instructions by scalar adds in the assembly restores bitwise replay,
#    s_nop padding around them does not (tools/probes/upcat_asm_variants.py, profiles/r04_nondeterminism.md).
#    tools/pk_opsel_audit.py counts such instructions in the built objects; tests/test_no_spills.py keeps the count at zero.
#  * gemm.hip: it pairs accumulator elements of DIFFERENT MFMA tiles for v_pk_* math, which needs register-to-register copies of
#    whole accumulator tiles — +64..128 VGPRs and scratch spills in the fused epilogues (round 1);
#  * the NMF wave programs: v_pk_fma_f32 / v_pk_mul_f32 pairs whose 64-bit register alignment costs more v_mov than the packing
#    saves: the rank-2 HALS backward of the generic-patch core went from 234 to 154 VGPRs and 43.9 -> 38.3 ms per cfg-5 step (round 3).
# Exception (SLP left on): gemm_bx.hip — without it the compile does not finish in five minutes, and the audit finds no such
# instruction in it.  (The standalone ft.NMF units nmf_r*.hip were an exception at first — round 3 believed they spill more
# without SLP; measured in round 4 they spill LESS: 34 kernels / 12.6 KB of scratch against 51 / 49 KB — and carried 17 000 of
# the op_sel'd instructions.)
_NO_SLP = ["-fno-slp-vectorize"]
_SLP_ON = {"gemm_bx.hip"}


class _PerFile(dict):
    def get(self, src, default=None):
        return [] if src in _SLP_ON else _NO_SLP


PER_FILE_FLAGS = _PerFile()


def _hipcc() -> str:
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: cannot build libfactorizer_hip.so")
    return exe


def sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _deps_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".inc", ".hip")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _flags_of(src):
    return " ".join([*FLAGS, *PER_FILE_FLAGS.get(src, [])])


def _stale_flags() -> bool:
    """an object compiled with other flags than today's (the .flags stamp beside it) makes the library stale"""
    for src in sources():
        stamp = os.path.join(OBJ, src[:-4] + ".o.flags")
        if not os.path.exists(stamp) or open(stamp).read() != _flags_of(src):
            return True
    return False


def needs_build() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < _deps_mtime() or _stale_flags()


def _hdr_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src: str) -> str:
    obj = os.path.join(OBJ, src[:-4] + ".o")
    flags = [*FLAGS, *PER_FILE_FLAGS.get(src, [])]
    stamp, want = obj + ".flags", " ".join(flags)   # an object built with other flags is stale, whatever its age
    fresh = os.path.exists(stamp) and open(stamp).read() == want
    if fresh and os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(CSRC, src)), _hdr_mtime()):
        return obj
    cmd = [_hipcc(), *flags, "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
    with open(stamp, "w") as f:
        f.write(want)
    return obj


def build(force: bool = False, verbose: bool = True, jobs: int | None = None) -> str:
    if not force and not needs_build():
        return LIB
    os.makedirs(OBJ, exist_ok=True)
    srcs = sources()
    jobs = jobs or min(len(srcs), os.cpu_count() or 4)
    if verbose:
        print(f"[factorizer_amd] hipcc {ARCH}: {len(srcs)} translation units, {jobs} jobs", flush=True)
    with cf.ThreadPoolExecutor(jobs) as ex:
        objs = list(ex.map(_compile, srcs))
    tmp = LIB + ".tmp"
    r = subprocess.run([_hipcc(), "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", tmp, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stderr[-4000:])
    os.replace(tmp, LIB)
    if verbose:
        print(f"[factorizer_amd] built {LIB}", flush=True)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
