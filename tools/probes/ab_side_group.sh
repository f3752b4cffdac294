# grouped weight-gradient launch of a block on the second stream (FZ_SIDE_WGRAD=<max voxel columns>) vs on the main stream (0)
for G in 0 100000 600000 0 100000 600000; do
FZ_SIDE_WGRAD=$G python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('side',$G,'ms_per_step',d['ms_per_step'], {s:r['by_stage'][s]['kernel_ms'] for s in ('stage0','stage1','stage2-4')})"
done
