mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_dense.py tests/test_gpu_bx.py tests/test_gpu_bf16.py tests/test_gpu_model.py -x -q 2>&1 | tail -4
for i in 1 2; do python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_bwd_wgrad_32','mlp_chain_fwd_32','mlp_chain_fwd_64','mlp_chain_bwd_64','act_linear_res_32->32','nmf_cf_bwd_32x128x128x128')})"; done
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('bf16 ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_bwd_wgrad_32','mlp_chain_fwd_32')})"
