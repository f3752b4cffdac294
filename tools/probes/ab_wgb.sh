# mlp_chain_bwd_wgrad_32 with its weight-gradient passes on v_mfma_f32_16x16x4_f32 (FZ_CHAIN_WGB=0) or on the bf16 pipe from two-level
# hi / lo planes in LDS (1, the shipped form): parity tests, then per-kernel ms from bench.py (fp32 and bf16), same box, probe library
# (python tools/probes/build_alt.py tools/probes/bin/lib_probe.so -DFZ_PROBE gemm.hip)
mkdir -p gpurun_out/r06
export FZ_LIB_PATH=tools/probes/bin/lib_probe.so
for L in 0 1 0 1; do
  echo "== FZ_CHAIN_WGB=$L"
  FZ_CHAIN_WGB=$L python -m pytest tests/test_gpu_dense.py -q -k "mlp_chain_backward_with_weight_gradients" 2>&1 | tail -2
  FZ_CHAIN_WGB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('f32 ms_per_step',d['ms_per_step'], 'mlp_chain_bwd_wgrad_32', k['mlp_chain_bwd_wgrad_32'])"
  FZ_CHAIN_WGB=$L python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs --dtype bf16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('bf16 ms_per_step',d['ms_per_step'], 'mlp_chain_bwd_wgrad_32', k['mlp_chain_bwd_wgrad_32'])"
done
