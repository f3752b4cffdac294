# BASELINE configs[4] step with the later windows of the generic-patch backward as read-modify-write (FZ_PCF_SEPARATE=0) or into their own buffer + fz_act_add; then the NMF rows of the step
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -3
for SEP in 0 1; do
  FZ_PCF_SEPARATE=$SEP python - <<PY
import sys, json, os
sys.path.insert(0, "tools")
import bench_configs as BC
print("FZ_PCF_SEPARATE", os.environ["FZ_PCF_SEPARATE"])
BC.cfg5(batches=(4,), dtypes=("bf16", "f32"))
PY
done
python tools/probes/cfg5_kernel_table.py
