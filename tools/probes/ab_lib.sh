# same-box A/B of two builds of the library (tools/probes/ablib/old.so, new.so) through FZ_LIB_PATH
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cf2.py tests/test_gpu_model.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -2
for L in old new old new; do FZ_LIB_PATH=$PWD/tools/probes/ablib/$L.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('$L ms_per_step',d['ms_per_step'], {n:k[n] for n in ('nmf_cf_bwd_32x128x128x128','nmf_cf_fwd_32x128x128x128','nmf_cf_bwd_64x64x64x64','nmf_cf_fwd_64x64x64x64','nmf_cf_bwd_128x32x32x32')}, 'frac', d['roofline']['frac'])"; done
for L in old new; do FZ_LIB_PATH=$PWD/tools/probes/ablib/$L.so python bench.py --steps 10 --warmup 3 --no-cpu-baseline --dtype bf16 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('bf16 $L ms_per_step',d['ms_per_step'], {n:k[n] for n in ('nmf_cf_bwd_32x128x128x128','nmf_cf_fwd_32x128x128x128')})"; done
