# grouped weight-gradient launches: partial-sum units per problem (FZ_WGRAD_UNITS) against step time and the wgrad rows of bench.py
for u in 1024 512 256 2048 1024; do
  FZ_WGRAD_UNITS=$u python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); t=d['roofline']['native_kernels_ms_per_step']
print($u, d['ms_per_step'], {k:v for k,v in t.items() if k.startswith('wgrad_block') or k.startswith('wgrad_conv_k2') or k.startswith('wgrad_tconv') or k.startswith('wgrad_linear') or k.startswith('wgrad_cat')})"
done
