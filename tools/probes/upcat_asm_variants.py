"""Assembly-level bisect of the eight-wave upcat reproducer (tools/probes/upcat_r03.hip, -DUPCAT_WAVES=8): which instruction
of the SLP-vectorised epilogue makes the kernel stop replaying bit for bit?

  python tools/probes/upcat_asm_variants.py build     here: device assembly -> edited variants -> code objects -> libraries
  python tools/probes/upcat_asm_variants.py run       on a GPU box: tools/probes/upcat_check.py against every variant

Flow per variant: hipcc -S (device only) -> regex edit of the fp32 kernel's text -> clang -x assembler -> ld.lld -> offload
bundle -> host object with -fcuda-include-gpubinary -> link with the default build's other objects."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
CL = "/opt/rocm/lib/llvm/bin"
OUT = os.path.join(ROOT, "tools/probes/bin")
SRC = os.path.join(ROOT, "tools/probes/upcat_r03.hip")
FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wno-comment", "-ffp-contract=fast", "-I" + os.path.join(ROOT, "factorizer_amd/csrc"),
         "-DUPCAT_WAVES=8"]
KERNEL = "_ZN2fz15upcat_bx_kernelIfEEvNS_10UpcatArgsTIT_EEl"

PK = re.compile(r"^\tv_pk_add_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\] (op_sel:\[0,1\]|op_sel_hi:\[1,0\])\s*$")


def scalar_adds(m):
    d0, d1, a0, a1, b0, b1 = (int(m.group(i)) for i in range(1, 7))
    b = b1 if m.group(7).startswith("op_sel:") else b0      # both halves add the same bias register
    lo, hi = f"\tv_add_f32_e32 v{d0}, v{a0}, v{b}", f"\tv_add_f32_e32 v{d1}, v{a1}, v{b}"
    assert d1 not in (a0, b) or d0 not in (a1, b)
    return [hi, lo] if d0 in (a1, b) else [lo, hi]


def edit(lines, variant):
    out, inside = [], False
    for ln in lines:
        if ln.startswith(KERNEL + ":"):
            inside = True
        if inside and "s_endpgm" in ln:
            inside = False
        m = PK.match(ln) if inside else None
        if m:
            sel = m.group(7).startswith("op_sel:")
            if variant == "nosel" and sel:
                out += scalar_adds(m); continue
            if variant == "nohi" and not sel:
                out += scalar_adds(m); continue
            if variant == "nopk":
                out += scalar_adds(m); continue
            if variant == "nop_sel" and sel:
                out += ["\ts_nop 7", ln, "\ts_nop 7"]; continue
        out.append(ln)
    return out


VARIANTS = ["base", "nosel", "nohi", "nopk", "nop_sel"]


def sh(*cmd):
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode:
        sys.exit(" ".join(cmd) + "\n" + r.stderr[-3000:])


def build():
    sys.path.insert(0, ROOT)
    from factorizer_amd import build as B
    B.build(verbose=False)
    os.makedirs(OUT, exist_ok=True)
    tmp = os.path.join(OUT, "asm")
    os.makedirs(tmp, exist_ok=True)
    dev = os.path.join(tmp, "dev.s")
    sh("hipcc", "--offload-arch=gfx950", *FLAGS, "--cuda-device-only", "-S", SRC, "-o", dev)
    lines = open(dev).read().split("\n")
    assert any(l.startswith(KERNEL + ":") for l in lines), "kernel symbol not found"
    others = [os.path.join(B.OBJ, f) for f in os.listdir(B.OBJ) if f.endswith(".o") and f != "upcat.o"]
    for v in VARIANTS:
        s = os.path.join(tmp, f"dev_{v}.s")
        ed = edit(lines, v)
        open(s, "w").write("\n".join(ed))
        n_pk = sum(1 for l in ed if "v_pk_add_f32" in l and "op_sel" in l)
        sh(CL + "/clang", "-x", "assembler", "-target", "amdgcn-amd-amdhsa", "-mcpu=gfx950", "-c", s, "-o", s[:-2] + ".o")
        sh(CL + "/ld.lld", "-shared", s[:-2] + ".o", "-o", s[:-2] + ".hsaco")
        sh(CL + "/clang-offload-bundler", "-type=o", "-bundle-align=4096", "-targets=host-x86_64-unknown-linux-gnu,hipv4-amdgcn-amd-amdhsa--gfx950",
           "-input=/dev/null", "-input=" + s[:-2] + ".hsaco", "-output=" + s[:-2] + ".hipfb")
        sh("hipcc", "--offload-arch=gfx950", *FLAGS, "--cuda-host-only", "-Xclang", "-fcuda-include-gpubinary", "-Xclang", s[:-2] + ".hipfb", "-c", SRC,
           "-o", s[:-2] + "_host.o")
        sh("hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", os.path.join(OUT, f"libfz_upcat_asm_{v}.so"), *others, s[:-2] + "_host.o")
        print(f"built {v}: {n_pk} op_sel'd v_pk_add_f32 left in the module text")


def run():
    for v in VARIANTS:
        for rep in range(2):
            env = dict(os.environ, FZ_LIB_PATH=os.path.join(OUT, f"libfz_upcat_asm_{v}.so"))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "tools/probes/upcat_check.py")], env=env, capture_output=True, text=True, cwd=ROOT)
            first = [l for l in r.stdout.splitlines() if l.startswith("replay equal")]
            bad = [l for l in r.stdout.splitlines() if l.startswith("bad elements")]
            print(f"{v:8s} run {rep}: {first[0] if first else r.stderr[-300:]} | {bad[0][:60] if bad else ''}", flush=True)


if __name__ == "__main__":
    build() if sys.argv[1] == "build" else run()
