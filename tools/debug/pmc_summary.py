"""Per-kernel averages of rocprofv3 --pmc counter_collection.csv files (several passes may be given)."""
import collections, csv, re, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip()[:64]
        if not name.startswith("fz::"):
            continue
        agg[(name, int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for (name, grid), cs in sorted(agg.items()):
    line = " ".join(f"{c}={sum(v)/len(v):.3g}" for c, v in sorted(cs.items()))
    print(f"{name} grid={grid} n={len(next(iter(cs.values())))}: {line}")
