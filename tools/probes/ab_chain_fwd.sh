# forward MLP chain at C = 32 on fp32 MFMAs (FZ_CHAIN_FWD_BX=0) vs split-bf16 products: tests, then per-kernel ms from bench.py, fp32 and bf16
python -m pytest tests/test_gpu_dense.py tests/test_gpu_bx.py tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q 2>&1 | tail -2
for v in 0 2 1 2 1; do FZ_CHAIN_FWD_BX=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('FZ_CHAIN_FWD_BX=$v ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_fwd_32','mlp_chain_bwd_wgrad_32','ln_linear_32->32')})"; done
for v in 0 1; do FZ_CHAIN_FWD_BX=$v python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import sys,json,os
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('bf16 FZ_CHAIN_FWD_BX=$v ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_fwd_32',)})"; done
