#!/bin/bash
# Is the second stream (weight gradients of the deep stages beside the main stream's kernels) bitwise reproducible with the
# round-4 build (-fno-slp-vectorize: no op_sel'd packed-fp32 instructions)?  Round 3 saw one NMF matrix differ per bf16 step.
mkdir -p gpurun_out/r04
for lim in 0 300000 100000000; do
  echo "== FZ_SIDE_WGRAD=$lim step_replay (README size, fp32 + bf16, 8 replays)"
  FZ_SIDE_WGRAD=$lim python tools/probes/step_replay.py 8 2>&1 | grep -v amdgpu.ids
done
echo "== cfg5 full size (160x192x160, fp32 + bf16 replay asserts), FZ_SIDE_WGRAD=100000000"
FZ_SIDE_WGRAD=100000000 python -m pytest tests/test_gpu_cfg5.py -q -m gpu -x 2>&1 | tail -3
echo "== concurrent core backward probe"
python tools/probes/core_concurrent2.py 2>&1 | grep -v amdgpu.ids
for lim in 0 300000; do
  echo "== bench FZ_SIDE_WGRAD=$lim"
  FZ_SIDE_WGRAD=$lim python bench.py --steps 20 --warmup 4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'])"
done
