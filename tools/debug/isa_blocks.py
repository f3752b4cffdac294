"""Disassemble one object of the library and print, per kernel matching a substring, the basic
blocks that contain MFMAs: instruction count, MFMA count and the s_waitcnt sequence.

    python tools/debug/isa_blocks.py gemm gemm_stream_kernel
"""
import re
import subprocess
import sys

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    unit, pat = sys.argv[1], sys.argv[2]
    src = f"factorizer_amd/csrc/{unit}.hip"
    tmp = f"/tmp/isa_{unit}.s"
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-S", "--cuda-device-only",
                           "-o", tmp, src])
    cur, blocks, name = None, [], None
    for line in open(tmp):
        s = line.strip()
        m = re.match(r"^(_Z\w+):", s)
        if m:
            name = m.group(1)
            cur = []
            blocks.append((name, cur))
            continue
        if re.match(r"^\.LBB\d+_\d+:", s):
            cur = []
            blocks.append((name, cur))
            continue
        if cur is not None and s and not s.startswith((".", ";")):
            cur.append(s)
    for name, b in blocks:
        if name is None or pat not in name:
            continue
        nm = sum("v_mfma" in i for i in b)
        if nm == 0:
            continue
        waits = [re.sub(r"\s+", "", i.split("s_waitcnt")[1]) for i in b if i.startswith("s_waitcnt")]
        tag = re.search(r"ILi.*?EEv", name)
        print(f"{tag.group(0) if tag else name[:60]}: instrs {len(b)} mfma {nm} waits {len(waits)} {waits[:24]}")


if __name__ == "__main__":
    main()
