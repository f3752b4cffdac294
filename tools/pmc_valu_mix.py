"""Dynamic VALU instruction mix per kernel from rocprofv3 --pmc passes with the SQ_INSTS_VALU_* type counters (counter_collection.csv).
For every fz:: kernel the launches with the LARGEST grid are averaged per counter.  SQ_INSTS_VALU counts wave instructions; the typed
counters partition the floating-point part (FMA / ADD / MUL / TRANS), INT32 and CVT are listed when the device exposes them; the rest
(moves, selects, compares, bit operations, DPP movement) is the difference.  usage: python tools/pmc_valu_mix.py pass.csv [pass2.csv]"""
import collections
import csv
import re
import sys


def main():
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[1:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if name.startswith("_ZN2fz"):      # rocprofv3 leaves names with bf16 template arguments mangled: _ZN2fz<len><name>I...
                m = re.match(r"_ZN2fz(\d+)", name)
                n = int(m.group(1))
                name = "fz::" + name[m.end():m.end() + n] + ("<bf16 storage>" if "DF16b" in name else "")
            else:
                name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
                name = re.sub(r"<.*", "", name)
            if not name.startswith("fz::"):
                continue
            agg[name][(int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    print("| kernel (largest grid) | waves' VALU instr | FMA f32 | MUL f32 | ADD f32 | TRANS | INT32 | CVT | other (mov / select / cmp / bits / dpp) | FMA share |")
    print("|---|---|---|---|---|---|---|---|---|---|")
    rows = []
    for name, d in agg.items():
        big = max(g for g, _ in d)
        c = {cn: sum(v) / len(v) for (g, cn), v in d.items() if g == big}
        tot = c.get("SQ_INSTS_VALU", 0)
        if tot <= 0:
            continue
        rows.append((tot, name, c))
    for tot, name, c in sorted(rows, reverse=True)[:16]:
        g = lambda k: c.get("SQ_INSTS_VALU_" + k, 0.0)  # noqa: E731
        known = g("FMA_F32") + g("MUL_F32") + g("ADD_F32") + g("TRANS_F32") + g("INT32") + g("CVT")
        print(f"| {name} | {tot:.3e} | {g('FMA_F32') / tot:.3f} | {g('MUL_F32') / tot:.3f} | {g('ADD_F32') / tot:.3f} | {g('TRANS_F32') / tot:.3f} | "
              f"{g('INT32') / tot:.3f} | {g('CVT') / tot:.3f} | {max(tot - known, 0) / tot:.3f} | {g('FMA_F32') / tot:.3f} |")


if __name__ == "__main__":
    main()
