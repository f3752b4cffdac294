"""One training step out of a rocprofv3 kernel trace: per-kernel start / duration / stream, and busy time per stream.
usage: python tools/step_timeline.py <kernel_trace.csv> [--full]"""
import csv, re, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
a, b = idx[-2] + 1, idx[-1] + 1
step = sorted(rows[a:b], key=lambda r: int(r['Start_Timestamp']))
t0 = int(step[0]['Start_Timestamp'])
span = (max(int(r['End_Timestamp']) for r in step) - t0) / 1e6
busy = collections.Counter(); cnt = collections.Counter()
agg = collections.Counter(); aggn = collections.Counter()
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    busy[r['Queue_Id']] += d; cnt[r['Queue_Id']] += 1
    n = re.sub(r'^void ', '', r['Kernel_Name']).replace('fz::', '')
    n = re.sub(r'\(.*', '', n)
    g = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    agg[(n[:60], g, r['Queue_Id'])] += d; aggn[(n[:60], g, r['Queue_Id'])] += 1
    if '--full' in sys.argv:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e3:9.1f} {d:8.1f} q{r['Queue_Id']} g{g:>8} v{r['VGPR_Count']:>3} lds{r['LDS_Block_Size']:>6} {n[:70]}")
print(f"step span {span:.3f} ms, {len(step)} launches; busy per queue (ms): " + ", ".join(f"q{q}: {v/1e3:.2f} ({cnt[q]})" for q, v in busy.items()))
for (n, g, q), v in sorted(agg.items(), key=lambda kv: -kv[1])[:45]:
    print(f"{v:8.1f} us  x{aggn[(n,g,q)]:<2} q{q} g{g:>8}  {n}")
