"""Assemble profiles/<name>.md from a rocprofv3 kernel-trace/stats run, the PMC summary and a bench line.
usage: python tools/make_profile_md.py <prof_dir/prefix> <pmc_traffic.json> <bench.json> <title> <notes> [<tag>] > profiles/xxx.md
(<tag> = prefix of the sibling files named in the footer, default r02_p2)"""
import collections
import csv
import json
import subprocess
import sys


def main():
    prefix, pmc_path, bench_path, title, notes = sys.argv[1:6]
    tag = sys.argv[6] if len(sys.argv) > 6 else "r02_p2"
    tab = subprocess.run([sys.executable, "tools/prof_summary.py", prefix + "_kernel_stats.csv", "30"],
                         capture_output=True, text=True, check=True).stdout
    rows = list(csv.DictReader(open(prefix + "_kernel_trace.csv")))
    agg = collections.defaultdict(list)
    for r in rows:
        n = r["Kernel_Name"]
        if any(s in n for s in ("nmf_cf_bwd_tile", "nmf_cf_bwd_gram", "nmf_cf_fwd_tile", "gemm_chain", "upcat_bx", "conv3_fwd_bx", "gemm_p32", "gn_fwd", "gemm_resident", "gemm_dw_kernel")):
            key = (n.split("(")[0].replace("void ", ""), int(r["Grid_Size_X"]))
            agg[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    pmc = json.load(open(pmc_path))
    b = json.load(open(bench_path))
    r = b["roofline"]
    out = [f"# {title}\n",
           "`rocprofv3 --kernel-trace --stats --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline` "
           "(`tools/profile_round.sh`: 5 steps under the profiler; summary by `tools/prof_summary.py`).\n", notes + "\n", tab,
           "\n## Launch duration by grid size (kernel trace) — the stage-0 launches are the ones `bench.py` reports\n",
           "| kernel | grid (threads) | launches | avg us |\n|---|---|---|---|"]
    for k, v in sorted(agg.items()):
        out.append(f"| `{k[0]}` | {k[1]} | {len(v)} | {sum(v) / len(v):.1f} |")
    unit = r["unit"]
    head = (f"{r['achieved']:.1f} {unit} = {r['frac']:.3f} of the {r['peak']:.1f} {unit} fp32 matrix peak — the binding roof at "
            f"{r.get('flop_per_algorithmic_byte', 0):.1f} flop per algorithmic byte; {r.get('hbm_GBps', 0):.0f} GB/s of algorithmic bytes = "
            f"{r.get('hbm_frac', 0):.3f} of 8 TB/s" if r["bound"] == "mfma" else
            f"{r['achieved']:.0f} GB/s of algorithmic bytes = {r['frac']:.3f} of 8 TB/s")
    out.append(f"\n`bench.py` (HIP events on the launch stream around the dominant kernel only, default run on the same "
               f"box): {b['value']} {b['unit']}, {b['ms_per_step']} ms/step; dominant kernel `{r['kernel']}`: avg launch "
               f"{r['avg_launch_ms'] * 1e3:.1f} us, {r['algorithmic_bytes_per_launch'] / 1e9:.3f} GB algorithmic per launch, {head}.\n")
    out.append("## HBM traffic per launch from PMC counters (separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes, "
               "FETCH_SIZE x2 per the gfx950 correction; `tools/pmc_traffic.py` -> `profiles/" + tag.split("_")[0] + "_pmc_traffic.json`; stage-0 launches)\n")
    out.append("| kernel | fetch (corrected) MB | write MB | traffic MB | algorithmic MB |\n|---|---|---|---|---|")
    r5 = tag.startswith(("r05", "r06"))   # round 5 on: window 0 of the backward runs the row-space kernel, window 1 the general one
    alg = {"fz::nmf_cf_bwd_gram_kernel<4, false, float, 0>": (3 * 536.87, " (window 0: t, dL/da in; dL/dt out)"),
           "fz::nmf_cf_bwd_tile_kernel<1, 1, 4, false, float>": ((4 if r5 else 3.5) * 536.87, " (window 1: + the running sum)" if r5 else " (avg over the 2 windows)"),
           "fz::nmf_cf_fwd_tile_kernel<1, 1, 8, false, float>": (2.5 * 536.87, " (avg over the 2 windows)"),
           "fz::gemm_chain_kernel<true, 2, 2, float>": (7 * 536.87, ""), "fz::gemm_chain_kernel<false, 2, 2, float>": (4 * 536.87, ""),
           "fz::gemm_chain_kernel<false, 2, 2, float, true, true>": (6 * 536.87 + 25.2, " (round 5: out_proj + residual + MLP chain, a, x in; x1, z1 (2), y out; + the head's logits on one of the two launches)"),
           "fz::upcat_bx_kernel<float, true>": (3.5 * 536.87 + 16.8, " (round 5: skip, deep in; out, t, statistics out)"),
           "fz::conv3_fwd_bx_kernel<1, 2, float, true>": (2 * 536.87 + 67.1 + 16.8, " (round 5: image in; x, t, statistics out)"),
           "fz::gemm_chain_bwd_wg_kernel<float>": (5 * 536.87, " (round 4: the residual rows g2 come from LDS; rounds 2-3 read them a second time, +537)"),
           "fz::gemm_chain_bwd_wg_kernel<float, 1, 0, true>": (5 * 536.87, " (round 4: the residual rows g2 come from LDS; rounds 2-3 read them a second time, +537)"),
           "fz::gemm_dw_kernel<true, float>": (4 * 536.87, ""), "fz::gemm_dw_kernel<false, float>": (3 * 536.87, "")}
    for k, (a, note) in alg.items():
        if k in pmc:
            v = pmc[k]
            out.append(f"| `{k}` | {v['fetch_bytes_corrected'] / 1e6:.0f} | {v['write_bytes'] / 1e6:.0f} | "
                       f"{v['traffic_bytes'] / 1e6:.0f} | {a:.0f}{note} |")
    out.append(f"\nFull bench line of that run: `profiles/" + tag.split("_")[0] + "_bench_n1.json`; SQ counters (VALU / MFMA / LDS / wait shares) of the same "
               f"command: `profiles/{tag}_pmc_sq.md`.\n")
    print("\n".join(out))


if __name__ == "__main__":
    main()
