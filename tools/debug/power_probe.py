"""Is the chip power/clock-limited when fp32 MFMAs run beside streaming traffic?  Samples rocm-smi while a GEMM loops."""
import os, subprocess, sys, threading, time, torch
sys.path.insert(0, '.')
import factorizer_amd._native as N
variant = sys.argv[1] if len(sys.argv) > 1 else "full"
if variant != "full":
    N.LIB_PATH = os.path.abspath(f"tools/debug/bin/libfz_probe_{variant}.so")
from factorizer_amd import pointwise as PW
DEV = 'cuda:0'
B, Cin, Cout, S = 16, 64, 128, 64
V = S ** 3
x = torch.randn(B, Cin, V, device=DEV); w = torch.randn(Cout, Cin, device=DEV); b = torch.randn(Cout, device=DEV)
y = torch.empty(B, Cout, V, device=DEV)
stop = False
samples = []
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=5).stdout
            samples.append(out.strip()[:600])
        except Exception as ex:
            samples.append(f"ERR {ex}")
        time.sleep(0.3)
th = threading.Thread(target=sampler); th.start()
t0 = time.perf_counter(); n = 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
while time.perf_counter() - t0 < 4.0:
    for _ in range(20):
        PW._gemm([x], w, y, B=B, Cin=Cin, Vin=V, M=Cout, K=Cin, Ncol=V, bias=b)
    n += 20
    torch.cuda.synchronize()
e.record(); torch.cuda.synchronize()
stop = True; th.join()
print(variant, "avg us per launch", s.elapsed_time(e) * 1e3 / n)
for smp in samples[2:6]:
    print(smp[:500])
