for G in 1 0 1 0; do
FZ_WGRAD_GROUP=$G python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; k=r['native_kernels_ms_per_step']
print('group',$G,'ms_per_step',d['ms_per_step'], {s:r['by_stage'][s]['kernel_ms'] for s in ('stage0','stage1','stage2-4')}, 'wgrad*', round(sum(v for n,v in k.items() if n.startswith('wgrad')),3))"
done
