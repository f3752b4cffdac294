# same-box A/B: default build vs tools/probes/ablib/<name>.so through FZ_LIB_PATH
N=${1:-maxilp}
for L in default $N default $N; do
  if [ $L = default ]; then unset FZ_LIB_PATH; else export FZ_LIB_PATH=$PWD/tools/probes/ablib/$L.so; fi
  python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['native_kernels_ms_per_step']
print('$L ms_per_step',d['ms_per_step'], {n:k[n] for n in ('mlp_chain_bwd_wgrad_32','mlp_chain_fwd_32','mlp_chain_fwd_64','mlp_chain_bwd_64','dgrad_wgrad_32','dgrad_lnbwd_wgrad_32','ln_linear_32->32','act_linear_res_32->32')})"
done
