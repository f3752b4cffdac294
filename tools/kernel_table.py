"""profiles/<name>.md: per-kernel table of one training step from a bench.py JSON line.
usage: python tools/kernel_table.py <bench.json or log> > profiles/r01_kernel_table.md"""
import json
import sys


def main():
    txt = open(sys.argv[1]).read().strip()
    b = json.loads(txt) if txt.startswith("{\n") or "\n" not in txt else json.loads(txt.splitlines()[-1])
    r = b["roofline"]
    k, g = r["native_kernels_ms_per_step"], r["native_kernels_GBps"]
    rows = sorted(k.items(), key=lambda kv: -kv[1])
    out = ["# Round 1 — per-kernel table of one training step (from `bench.py`, instrumented warm-up steps)\n",
           f"Default run: **{b['value']} {b['unit']}**, {b['ms_per_step']} ms/step; device copy on this box "
           f"{r.get('device_copy_GBps_this_box')} GB/s; HBM peak used for the fractions: 8000 GB/s.",
           "Timer keys = layer family + GEMM shape (`K->M` channels) or tensor shape; GB/s = algorithmic bytes of the "
           "launches (SURVEY §8d) / their duration, measured with HIP events around every native launch. The "
           "deep stages are MFMA- or latency-bound, not HBM-bound, so their GB/s column is only descriptive; "
           "deep-stage weight gradients overlap with the input-gradient chain on a second stream, so the column "
           "does not add up to the step time.\n",
           "| kernel / layer | ms per step | algorithmic GB/s | fraction of 8 TB/s |", "|---|---|---|---|"]
    for n, v in rows:
        if v >= 0.05:
            out.append(f"| `{n}` | {v:.3f} | {g[n]:.0f} | {g[n] / 8000:.2f} |")
    out.append(f"\n{len(rows)} timer keys, {sum(k.values()):.2f} ms summed.")
    print("\n".join(out))


if __name__ == "__main__":
    main()
