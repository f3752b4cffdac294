"""tabulate a gemm_bx_bench jsonl: python tools/probes/tab.py file.jsonl"""
import collections
import json
import sys
rows = [json.loads(l) for l in open(sys.argv[1])]
by = collections.OrderedDict()
vs = []
for r in rows:
    by.setdefault(r['shape'], {})[r['variant']] = r
    if r['variant'] not in vs:
        vs.append(r['variant'])
print(f"{'shape':24s}" + ''.join(f"{v.replace('bx_',''):>8s}" for v in vs) + '   best: TB/s TF')
for s, d in by.items():
    best = min((d[v]['us'], v) for v in d if v != 'f32mfma')
    print(f"{s:24s}" + ''.join(f"{d[v]['us']:8.1f}" if v in d else f"{'-':>8s}" for v in vs) + f"   {d[best[1]]['TBps']:5.2f} {d[best[1]]['TFLOPs']:6.1f} {best[1]}")
