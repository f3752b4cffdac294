mkdir -p gpurun_out/r04
python - <<'PY' > gpurun_out/r04/cf2_time2.jsonl 2> gpurun_out/r04/cf2_time2.err
import os, sys, json, subprocess
# each configuration in its own process (knobs are read once)
for nowait in ("0", "1"):
    env = dict(os.environ, FZ_CF2_NOWAIT=nowait)
    r = subprocess.run([sys.executable, "tools/probes/cf2_time.py", "short"], env=env, capture_output=True, text=True)
    for l in r.stdout.splitlines():
        d = json.loads(l); d["nowait"] = int(nowait); print(json.dumps(d), flush=True)
    sys.stderr.write(r.stderr[-500:])
PY
