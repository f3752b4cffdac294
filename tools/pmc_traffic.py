"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs into per-kernel HBM traffic per launch.

usage: python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> > profiles/xxx.json

Correction per /opt/skills/guides/MI355X_MICROARCH.md §HBM: on gfx950 FETCH_SIZE (KB) reports
half of the bytes of a wide (16 B/lane) coalesced read → doubled here; WRITE_SIZE is exact for
16 B/lane streaming stores.  Both are L2-fabric-side counters (Infinity-Cache hits included).
For every kernel the launches with the largest grid are averaged (the stage-0 / BASELINE-size
launches)."""
import collections
import csv
import json
import re
import sys


def plain_name(name):
    """rocprofv3 leaves names with a __bf16 template argument mangled (`_ZN2fz22nmf_cf_bwd_tile_kernelILi1E...DF16bEEv...`):
    turn them into `fz::nmf_cf_bwd_tile_kernel<mangled:Li1E...DF16b>` so that the bf16 instantiations are listed like the others"""
    m = re.match(r"_ZN2fz(\d+)", name)
    if not m:
        return name
    n = int(m.group(1))
    ident = name[m.end():m.end() + n]
    rest = name[m.end() + n:]
    targs = rest[1:rest.index("EEv")] if rest.startswith("I") and "EEv" in rest else ""
    return f"fz::{ident}<mangled:{targs}>" if targs else f"fz::{ident}"


def load(path, cname):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != cname:
            continue
        name = plain_name(re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").strip())
        d[name].append((int(r["Grid_Size"]), float(r["Counter_Value"])))
    return d


def main():
    f = load(sys.argv[1], "FETCH_SIZE")
    w = load(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name in sorted(f):
        if not name.startswith("fz::"):
            continue
        big = max(g for g, _ in f[name])
        fv = [v for g, v in f[name] if g == big]
        wv = [v for g, v in w.get(name, []) if g == big]
        fetch = 2.0 * 1024.0 * sum(fv) / len(fv)
        write = 1024.0 * sum(wv) / len(wv) if wv else 0.0
        out[name] = {"launches_averaged": len(fv), "grid": big, "fetch_bytes_corrected": round(fetch),
                     "write_bytes": round(write), "traffic_bytes": round(fetch + write)}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
