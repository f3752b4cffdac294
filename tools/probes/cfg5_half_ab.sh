# BASELINE configs[4] ("cfg 5") training step with the 8 x 150 NMF kernels one matrix per wave (FZ_PCF_HALF=0) and two per wave
mkdir -p gpurun_out/r04
for H in 0 1; do
  FZ_PCF_HALF=$H python - <<PY
import sys, json, os
sys.path.insert(0, "tools")
import bench_configs as BC
print("FZ_PCF_HALF", os.environ["FZ_PCF_HALF"])
BC.cfg5(batches=(4,), dtypes=("bf16", "f32"))
PY
done
