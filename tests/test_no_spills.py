"""Register-allocation guard for the hot kernels (no GPU needed: hipcc cross-compiles gfx950).  A spilled VGPR is
scratch traffic inside the tile loop: 32 of them cost the fused MLP backward 10 % (935 -> 1025 us, +215 MB of writes in
the WRITE_SIZE counter) after a template change that looked harmless.  The kernels named here must stay spill-free."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from factorizer_amd import build as B

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "factorizer_amd", "csrc")

HOT = {
    "gemm.hip": ["gemm_chain_bwd_wg_kernelIfLi1ELi0E", "gemm_chain_bwd_wg_kernelIfLi2ELi0E", "gemm_chain_bwd_wg_kernelIfLi2ELi1E",
                 "gemm_dw_kernelILb1Ef", "gemm_dw_kernelILb0Ef", "gemm_chain64_kernelILb0Ef", "gemm_chain64_kernelILb1Ef",
                 "gemm_chain_kernelILb0ELi2ELi2Ef"],
    "nmf_cf.hip": ["nmf_cf_bwd_tile_kernelILi1ELi1ELi4ELb0Ef", "nmf_cf_fwd_tile_kernelILi1ELi1ELi8ELb0Ef"],
}


def _asm(src):
    with tempfile.NamedTemporaryFile(suffix=".s") as tmp:
        cmd = [B._hipcc(), *B.FLAGS, *B.PER_FILE_FLAGS.get(src, []), "-I", os.path.join(ROOT, "include"), "--cuda-device-only",
               "-S", os.path.join(CSRC, src), "-o", tmp.name]
        r = subprocess.run(cmd, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
        return open(tmp.name).read()


@pytest.fixture(scope="module")
def metadata():
    with ThreadPoolExecutor(len(HOT)) as ex:
        texts = dict(zip(HOT, ex.map(_asm, HOT)))
    out = {}
    for src, txt in texts.items():
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt):
            out[m.group(1)] = (int(m.group(2)), int(m.group(3)))
    return out


@pytest.mark.parametrize("src,frag", [(s, f) for s, fs in HOT.items() for f in fs])
def test_hot_kernel_does_not_spill(metadata, src, frag):
    hits = {k: v for k, v in metadata.items() if frag in k}
    assert hits, f"no kernel matching {frag} in {src}"
    for name, (vgprs, spills) in hits.items():
        assert spills == 0, f"{name}: {spills} spilled VGPRs ({vgprs} allocated)"
