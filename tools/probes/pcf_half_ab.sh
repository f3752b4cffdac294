# 8 x 150 NMF kernels one matrix per wave (FZ_PCF_HALF=0) vs two per wave: isolated launch times, then the parity tests
mkdir -p gpurun_out/r04
FZ_PCF_HALF=0 python tools/probes/pcf_half_time.py 2
python tools/probes/pcf_half_time.py 2
python -m pytest tests/test_gpu_parity.py tests/test_gpu_cfg5.py tests/test_gpu_bf16.py -x -q 2>&1 | tail -4
