"""Which launches of a rocprofv3 kernel trace of `bench.py` fall INSIDE training steps, and which are set-up?
A step ends with the single AdamW launch (fz::adamw_kernel); the window of step k is (end of adamw k-1, end of adamw k].  Everything
before the first window's start is set-up (parameter uploads, FlatAdamW moving every parameter into its flat buffer: one
device-to-device copy per parameter, ...).  Prints per-step counts of __amd_rocclr_copyBuffer and of every kernel that is not
the library's (at::native::*, rccl, fills), and the same totals for the set-up part.
usage: python tools/copies_in_steps.py <kernel_trace.csv> > profiles/rNN_step_copies.md"""
import csv
import re
import sys
from collections import Counter, defaultdict


def short(n):
    n = re.sub(r"\(.*", "", n)
    n = re.sub(r"<.*", "", n)
    return n.replace("void ", "")[:70]


def main():
    rows = [r for r in csv.DictReader(open(sys.argv[1])) if r.get("Kind", "KERNEL_DISPATCH") == "KERNEL_DISPATCH"]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ends = [int(r["End_Timestamp"]) for r in rows if "adamw_kernel" in r["Kernel_Name"]]
    if len(ends) < 2:
        print("fewer than two fz::adamw_kernel launches in the trace: no step windows")
        return
    per = [defaultdict(lambda: [0, 0]) for _ in range(len(ends) - 1)]
    setup = defaultdict(lambda: [0, 0])
    tail = defaultdict(lambda: [0, 0])
    nlib = [0] * (len(ends) - 1)
    tlib = [0] * (len(ends) - 1)
    for r in rows:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = short(r["Kernel_Name"])
        lib = name.startswith("fz::") or "dice_ce" in name
        k = next((i for i in range(len(ends) - 1) if ends[i] < e <= ends[i + 1]), None)
        if k is None:
            d = setup if e <= ends[0] else tail
            if not lib:
                d[name][0] += 1
                d[name][1] += e - s
            continue
        if lib:
            nlib[k] += 1
            tlib[k] += e - s
        else:
            per[k][name][0] += 1
            per[k][name][1] += e - s
    print(f"trace: {sys.argv[1].split('/')[-1]}; {len(rows)} kernel dispatches, {len(ends)} AdamW launches -> {len(ends) - 1} complete step windows "
          "(window k = after AdamW k-1 up to and including AdamW k)\n")
    print("| step window | library launches | library ms | non-library launches | non-library ms | of which __amd_rocclr_copyBuffer |")
    print("|---|---|---|---|---|---|")
    for k in range(len(ends) - 1):
        n = sum(v[0] for v in per[k].values())
        t = sum(v[1] for v in per[k].values())
        c = per[k].get("__amd_rocclr_copyBuffer", [0, 0])
        print(f"| {k + 1} | {nlib[k]} | {tlib[k] / 1e6:.3f} | {n} | {t / 1e6:.3f} | {c[0]} ({c[1] / 1e3:.1f} us) |")
    last = per[-1]
    print("\nnon-library kernels of the last window:\n")
    print("| kernel | launches | us |")
    print("|---|---|---|")
    for name, (n, t) in sorted(last.items(), key=lambda kv: -kv[1][1]):
        print(f"| {name} | {n} | {t / 1e3:.1f} |")
    print("\nbefore the first AdamW (set-up + step 1), non-library kernels:\n")
    print("| kernel | launches | us |")
    print("|---|---|---|")
    for name, (n, t) in sorted(setup.items(), key=lambda kv: -kv[1][1])[:12]:
        print(f"| {name} | {n} | {t / 1e3:.1f} |")
    if tail:
        print("\nafter the last AdamW (epilogue of the script), non-library kernels: " + ", ".join(f"{k} x{v[0]}" for k, v in tail.items()))


if __name__ == "__main__":
    main()
