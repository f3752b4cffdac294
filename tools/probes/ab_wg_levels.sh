# mlp_chain_bwd_wgrad_32 with its weight-gradient passes on fp32 MFMA (0) or on the bf16 pipe with 1 / 2 / 3 operand levels
mkdir -p gpurun_out/r04
for L in 0 2 3 1; do
  echo "== FZ_CHAIN_WG_LEVELS=$L"
  FZ_CHAIN_WG_LEVELS=$L python -m pytest tests/test_gpu_dense.py -q -k "mlp_chain_backward_with_weight_gradients" 2>&1 | tail -2
  FZ_CHAIN_WG_LEVELS=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('f32 ms_per_step',d['ms_per_step'], 'mlp_chain_bwd_wgrad_32', k['mlp_chain_bwd_wgrad_32'])"
  FZ_CHAIN_WG_LEVELS=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('bf16 ms_per_step',d['ms_per_step'], 'mlp_chain_bwd_wgrad_32', k['mlp_chain_bwd_wgrad_32'])"
done
