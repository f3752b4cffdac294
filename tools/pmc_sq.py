"""Per-kernel SQ counter summary of rocprofv3 --pmc passes (counter_collection.csv files) → JSON + markdown rows.

usage: python tools/pmc_sq.py out.json pass1_counter_collection.csv [pass2 ...]

For every fz:: kernel the launches with the LARGEST grid (the stage-0 / BASELINE-size launches) are
averaged per counter; derived ratios (units per /opt/skills/guides/MI355X_MICROARCH.md §cycle constants:
SQ_BUSY_CYCLES and SQ_*_BUSY_CYCLES count cycles summed over the shader engines / CUs they are collected
on; SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles per wave; ratios between counters of the same family are
unit-free):
  valu_util      = SQ_ACTIVE_INST_VALU / (SQ_ACTIVE_INST_ANY + SQ_WAIT_ANY + SQ_WAIT_INST_ANY)
                   share of a wave's lifetime in which it is executing a VALU instruction
  wait_share     = SQ_WAIT_ANY / (same denominator)      parked on s_waitcnt / barrier (memory, LDS, sync)
  issue_stall    = SQ_WAIT_INST_ANY / (same denominator) waiting to issue (pipe busy / dependency)
  mfma_busy      = (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs) / (SQ_BUSY_CYCLES / 32 shader engines)
                   busy fraction of one matrix pipe over the launch
  valu_issue_busy = 4 cycles x SQ_INSTS_VALU / 1024 SIMDs over the same denominator
  lds_conflict   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE    extra LDS cycles due to bank conflicts
  valu_per_lds   = SQ_INSTS_VALU / SQ_INSTS_LDS
"""
import collections
import csv
import json
import re
import sys


def main():
    out_path = sys.argv[1]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in sys.argv[2:]:
        for r in csv.DictReader(open(path)):
            name = r["Kernel_Name"]
            if name.startswith("_ZN2fz"):      # names with bf16 template arguments stay mangled in rocprofv3's output
                m = re.match(r"_ZN2fz(\d+)", name)
                name = "fz::" + name[m.end():m.end() + int(m.group(1))] + "<" + name[m.end() + int(m.group(1)):][:40] + ">"
            else:
                name = re.sub(r"\(.*", "", name).replace("void ", "").strip()
            if not name.startswith("fz::"):
                continue
            agg[name][(int(r["Grid_Size"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    out = {}
    for name, d in sorted(agg.items()):
        big = max(g for g, _ in d)
        c = {cn: sum(v) / len(v) for (g, cn), v in d.items() if g == big}
        n = max(len(v) for (g, cn), v in d.items() if g == big)
        life = c.get("SQ_ACTIVE_INST_ANY", 0) + c.get("SQ_WAIT_ANY", 0) + c.get("SQ_WAIT_INST_ANY", 0)
        der = {}
        if life > 0:
            der["valu_util"] = c.get("SQ_ACTIVE_INST_VALU", 0) / life
            der["wait_share"] = c.get("SQ_WAIT_ANY", 0) / life
            der["issue_stall"] = c.get("SQ_WAIT_INST_ANY", 0) / life
            if "SQ_ACTIVE_INST_LDS" in c:
                der["lds_inst_share"] = c["SQ_ACTIVE_INST_LDS"] / life
            if "SQ_ACTIVE_INST_VMEM" in c:
                der["vmem_inst_share"] = c["SQ_ACTIVE_INST_VMEM"] / life
        if c.get("SQ_BUSY_CYCLES", 0) > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
            # SQ_BUSY_CYCLES sums the 32 shader engines (calibrated on a 424 us launch: 3.08e7 / 32 = 0.96e6 cycles),
            # SQ_VALU_MFMA_BUSY_CYCLES sums the 1024 SIMDs (= 64 cycles x MFMA count for v_mfma_f32_32x32x2_f32):
            # busy fraction of ONE matrix pipe = (MFMA_BUSY / 1024) / (BUSY / 32)
            der["mfma_busy"] = c["SQ_VALU_MFMA_BUSY_CYCLES"] / c["SQ_BUSY_CYCLES"] / 32.0
            if "SQ_INSTS_VALU" in c:
                # VALU issue share of one SIMD: 4 cycles per wave64 instruction (MI355X_MICROARCH.md cycle constants)
                der["valu_issue_busy"] = 4.0 * c["SQ_INSTS_VALU"] / 1024.0 / (c["SQ_BUSY_CYCLES"] / 32.0)
        if c.get("SQ_LDS_IDX_ACTIVE", 0) > 0 and "SQ_LDS_BANK_CONFLICT" in c:
            der["lds_conflict"] = c["SQ_LDS_BANK_CONFLICT"] / c["SQ_LDS_IDX_ACTIVE"]
        if c.get("SQ_INSTS_LDS", 0) > 0 and "SQ_INSTS_VALU" in c:
            der["valu_per_lds"] = c["SQ_INSTS_VALU"] / c["SQ_INSTS_LDS"]
        out[name] = {"grid": big, "launches_averaged": n, "counters": {k: round(v, 1) for k, v in sorted(c.items())},
                     "derived": {k: round(v, 4) for k, v in der.items()}}
    json.dump(out, open(out_path, "w"), indent=1)
    print("| kernel (largest grid) | valu_util | wait_share | issue_stall | mfma_busy | valu_issue_busy | lds_conflict | VALU insts | MFMA insts |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, v in out.items():
        d, c = v["derived"], v["counters"]
        f = lambda k: (f"{d[k]:.3f}" if k in d else "–")  # noqa: E731
        print(f"| `{name[:70]}` {v['grid']} | {f('valu_util')} | {f('wait_share')} | {f('issue_stall')} | {f('mfma_busy')} | "
              f"{f('valu_issue_busy')} | {f('lds_conflict')} | {c.get('SQ_INSTS_VALU', 0):.3g} | {c.get('SQ_INSTS_MFMA', 0):.3g} |")


if __name__ == "__main__":
    main()
