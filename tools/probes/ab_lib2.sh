# same-box A/B: tools/probes/ablib/old.so vs new.so through FZ_LIB_PATH (build both with tools/probes/build_alt.py or by hand)
for L in old new old new; do
  FZ_LIB_PATH=$PWD/tools/probes/ablib/$L.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('$L ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_bwd_wgrad_32','mlp_chain_fwd_32')})"
done
FZ_LIB_PATH=$PWD/tools/probes/ablib/new.so python -m pytest tests/test_gpu_dense.py tests/test_gpu_model.py -x -q 2>&1 | tail -1
