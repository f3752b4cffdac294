mkdir -p gpurun_out/r02
run() { python bench.py --steps 10 --warmup 3 --no-cpu-baseline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print(d['ms_per_step'])"; }
python -c "import torch; print(torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
echo "base: $(run) $(run)"
echo "hp only: $(run --hp-stream) $(run --hp-stream)"
export FZ_SIDE_WGRAD=999999999
echo "defer all: $(run) $(run)"
echo "defer all + hp: $(run --hp-stream) $(run --hp-stream)"
export FZ_SIDE_WGRAD=0
echo "no side: $(run)"
