"""Condense a rocprofv3 *_kernel_stats.csv into a short table (kernel names shortened).
usage: python tools/prof_summary.py <kernel_stats.csv> [top_n] > profiles/xxx.md"""
import csv
import re
import sys


def short(name):
    name = re.sub(r"\(.*", "", name)
    name = re.sub(r"<.*", "", name)
    name = name.replace("void ", "")
    if name.startswith("Cijk_"):
        m = re.search(r"MT(\d+x\d+x\d+)", name)
        return "rocBLAS/Tensile " + name[:14] + (" MT" + m.group(1) if m else "")
    if name.startswith("_ZN2ck"):
        return "composable_kernel " + ("bwd_weight" if "bwd_weight" in name else "conv_fwd" if "conv_fwd" in name else "kernel")
    return name[:90]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
    agg = {}
    for r in rows:
        k = short(r["Name"])
        a = agg.setdefault(k, [0, 0.0])
        a[0] += int(r["Calls"])
        a[1] += float(r["TotalDurationNs"])
    tot = sum(v[1] for v in agg.values())
    print("| kernel | calls | total ms | avg us | % |")
    print("|---|---|---|---|---|")
    for k, (c, ns) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
        print(f"| {k} | {c} | {ns / 1e6:.3f} | {ns / c / 1e3:.1f} | {100 * ns / tot:.2f} |")
    print(f"\ntotal kernel time: {tot / 1e6:.3f} ms over {sum(v[0] for v in agg.values())} launches")


if __name__ == "__main__":
    main()
