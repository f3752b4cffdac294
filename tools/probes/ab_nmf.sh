python -m pytest tests/test_gpu_parity.py tests/test_gpu_cf2.py -x -q 2>&1 | tail -2
for i in 1 2 3; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('ms_per_step',d['ms_per_step'], {n:k[n] for n in ('nmf_cf_bwd_32x128x128x128','nmf_cf_fwd_32x128x128x128','nmf_cf_bwd_64x64x64x64','nmf_cf_fwd_64x64x64x64','mlp_chain_bwd_wgrad_32')})"; done
