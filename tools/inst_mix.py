"""Static instruction mix of built kernels: disassemble the gfx950 code object of one translation unit (factorizer_amd/csrc/build/
<tu>.o, no recompilation) and classify every instruction of the kernels whose demangled name contains <pattern>.
VALU classes: fma (v_fma_f32 / v_fmac / v_pk_fma_f32 / v_mad — a pk_fma counts as TWO lanes-worth of fp32 work), mul, add/sub,
max/min/cmp/cndmask (ReLU, gates, masks), cvt / pack (bf16 <-> fp32), mov / perm (register shuffles), dpp / readlane / permlane /
swizzle (cross-lane movement: the wave reductions), trans (rcp / exp / sqrt), int (address and index arithmetic).
A static count weights every instruction once — loops whose trip count is a run-time value (the T iterations) are not unrolled
in the listing — so the table is the composition of the program text; the DYNAMIC share comes from the SQ counters next to it.
usage: python tools/inst_mix.py <tu> <pattern> [<pattern> ...]"""
import os
import re
import subprocess
import sys
import tempfile
from collections import Counter

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "factorizer_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"


def classify(op, rest):
    cross = ("dpp" in rest or "row_" in rest or "quad_perm" in rest or op.startswith(("v_readlane", "v_readfirstlane", "v_permlane", "v_writelane", "ds_swizzle", "ds_bpermute", "ds_permute"))
             or op in ("v_mov_b32_dpp",))
    if op.startswith(("s_",)):
        return "salu/branch/wait"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("ds_"):
        return "cross-lane" if cross else "lds"
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if not op.startswith("v_"):
        return "other"
    base = op
    if cross and not op.startswith(("v_readlane", "v_readfirstlane", "v_permlane", "v_writelane")):
        # an arithmetic instruction with a DPP operand does its arithmetic AND the lane movement: counted as cross-lane+<class>
        pre = "dpp+"
    else:
        pre = ""
    if cross and pre == "":
        return "cross-lane"
    if re.match(r"v_(pk_)?(fma|fmac|mad|mac)_", base) or base.startswith("v_dot"):
        c = "fma"
    elif re.match(r"v_(pk_)?mul_(f32|f16|legacy)", base):
        c = "mul"
    elif re.match(r"v_(pk_)?(add|sub|subrev)_f", base):
        c = "add"
    elif re.match(r"v_(max|min|cmp|cmpx|cndmask|med3|pk_max|pk_min)", base):
        c = "max/cmp/select"
    elif base.startswith(("v_cvt", "v_pack", "v_lshl", "v_lshr", "v_and_", "v_or_", "v_bfi", "v_bfe", "v_perm", "v_alignb", "v_xor")):
        c = "cvt/pack/bits"
    elif base.startswith(("v_mov", "v_accvgpr", "v_swap")):
        c = "mov"
    elif base.startswith(("v_rcp", "v_rsq", "v_sqrt", "v_exp", "v_log", "v_sin", "v_cos")):
        c = "trans"
    elif re.match(r"v_(add|sub|mul|mad|lshl_add|add3|ashr|mul_lo|mul_hi|mul_u|mul_i|sub_|subrev_|addc|add_co|add_u|add_nc)", base):
        c = "int"
    else:
        c = "other-valu"
    return pre + c


def kernels(tu):
    obj = os.path.join(OBJ, tu + ".o")
    with tempfile.TemporaryDirectory() as td:
        fb, co = os.path.join(td, "fb.bin"), os.path.join(td, "co.elf")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fb}", obj], check=True, capture_output=True)
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fb}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True, capture_output=True)
        txt = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", co], check=True, capture_output=True, text=True).stdout
    out, cur = {}, None
    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            cur = m.group(1)
            out[cur] = []
            continue
        if cur is None:
            continue
        t = line.strip()
        if not t or t.startswith(("//", ";")):
            continue
        t = re.sub(r"//.*", "", t).strip()
        parts = t.split(None, 1)
        if parts and re.match(r"^[a-z_0-9]+$", parts[0]):
            out[cur].append((parts[0], parts[1] if len(parts) > 1 else ""))
    return out


def main():
    tu, pats = sys.argv[1], sys.argv[2:]
    ks = kernels(tu)
    names = list(ks)
    dem = subprocess.run(["c++filt"] + names, capture_output=True, text=True).stdout.strip().splitlines()
    for n, d in zip(names, dem):
        if not any(p in d or p in n for p in pats):
            continue
        ins = ks[n]
        cnt = Counter(classify(op, rest) for op, rest in ins)
        valu = {k: v for k, v in cnt.items() if k not in ("salu/branch/wait", "vmem", "lds", "mfma", "other")}
        nv = sum(valu.values())
        pk = sum(1 for op, _ in ins if op.startswith("v_pk_"))
        fma = sum(v for k, v in valu.items() if k.endswith("fma"))
        title = re.sub(r"\(.*", "", d)[:150]
        print(f"### `{title}`\n")
        print(f"{len(ins)} instructions in the text: {nv} VALU ({pk} of them packed fp32), {cnt['lds']} LDS, {cnt['vmem']} global, {cnt['salu/branch/wait']} scalar / branch / wait, {cnt['mfma']} MFMA\n")
        print("| VALU class | instructions | share of VALU |")
        print("|---|---|---|")
        for k, v in sorted(valu.items(), key=lambda kv: -kv[1]):
            print(f"| {k} | {v} | {100.0 * v / max(nv, 1):.1f} % |")
        print(f"\nFMA share of the VALU text: {100.0 * fma / max(nv, 1):.1f} %\n")


if __name__ == "__main__":
    main()
