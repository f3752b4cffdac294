# round 3: which test file running before tests/test_gpu_wide_nmf.py makes its per-call timing jump (the pause between calls: profiles/r03_wide_nmf_timing.json)
mkdir -p gpurun_out/r03
for f in test_gpu_bf16 test_gpu_bx test_gpu_deconver test_gpu_dense test_gpu_parity; do
  python -m pytest tests/$f.py tests/test_gpu_wide_nmf.py -q -m gpu -k "not fp64 or wide" 2>&1 | tail -2
  python - <<PY
import json
d=json.load(open("gpurun_out/parity.json"))
recs=d if isinstance(d,list) else d.get("records", d)
for r in (recs if isinstance(recs,list) else recs.values()):
    if isinstance(r,dict) and r.get("what")=="wide_nmf_16x262144_mu_r1_t5": print("$f", r["fwd_ms"], r["fwd_bwd_ms"])
PY
done
python -m pytest tests/test_gpu_cfg5.py -q -m gpu 2>&1 | tail -5
