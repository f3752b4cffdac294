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
# gemm.hip: the SLP vectorizer pairs accumulator elements of DIFFERENT MFMA tiles for v_pk_* math,
# which needs register-to-register copies of whole accumulator tiles (adjacent VGPRs hold the same
# column group, not the same row) — +64..128 VGPRs and scratch spills in the fused epilogues.
# -fno-slp-vectorize: the SLP vectorizer pairs elements of different MFMA tiles (gemm.hip) and, in the NMF wave programs, forms
# v_pk_fma_f32 / v_pk_mul_f32 pairs whose 64-bit register alignment costs more v_mov than the packing saves: the rank-2 HALS
# backward of the generic-patch core went from 234 to 154 VGPRs and 43.9 -> 38.3 ms per cfg-5 step without it (round 3)
_NO_SLP = ["-fno-slp-vectorize"]
PER_FILE_FLAGS = {"gemm.hip": _NO_SLP, "nmf_pcf.hip": _NO_SLP, "nmf_cf.hip": _NO_SLP}


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


def needs_build() -> bool:
    return not os.path.exists(LIB) or os.path.getmtime(LIB) < _deps_mtime()


def _hdr_mtime() -> float:
    m = 0.0
    for root in (CSRC, os.path.join(os.path.dirname(HERE), "include")):
        for f in os.listdir(root):
            if f.endswith((".h", ".inc")):
                m = max(m, os.path.getmtime(os.path.join(root, f)))
    return m


def _compile(src: str) -> str:
    obj = os.path.join(OBJ, src[:-4] + ".o")
    if os.path.exists(obj) and os.path.getmtime(obj) >= max(os.path.getmtime(os.path.join(CSRC, src)), _hdr_mtime()):
        return obj
    cmd = [_hipcc(), *FLAGS, *PER_FILE_FLAGS.get(src, []), "-c", os.path.join(CSRC, src), "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{r.stderr[-4000:]}")
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
