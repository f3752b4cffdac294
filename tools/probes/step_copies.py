"""Where do the ~48 device-to-device copies of a README-size training step (rocprof: __amd_rocclr_copyBuffer, 0.38 ms / step)
come from?  One step under torch.profiler with shapes and Python stacks; every aten::copy_ / clone / contiguous listed."""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench as Bn
import factorizer_amd as ft
from factorizer_amd.parallel import FlatGradSync
from torch.profiler import profile, ProfilerActivity

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = ft.Factorizer(**Bn.MODEL_KW).to(dev).train()
sync = FlatGradSync(model, num_buckets=4, overlap=True, late_wgrad_join=True)
opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
x = torch.rand(2, 4, 128, 128, 128, device=dev)
target = (torch.rand(2, 3, 128, 128, 128, device=dev) > 0.5).float()


def step():
    sync.zero_grad()
    loss = ft.dice_ce_loss(model(x), target)
    loss.backward()
    opt.step(grad_scale=sync.finish(average=False))


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
rows = {}
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::zero_", "aten::fill_", "aten::add_", "aten::add", "aten::mul",
                   "aten::sum", "aten::cat", "aten::_to_copy"):
        st = [s for s in (ev.stack or []) if "factorizer_amd" in s or "bench" in s or "probes" in s][:3]
        key = (ev.name, str(ev.input_shapes)[:80], " <- ".join(s.split("/")[-1] for s in st))
        r = rows.setdefault(key, [0, 0.0])
        r[0] += 1
        r[1] += ev.device_time_total if hasattr(ev, "device_time_total") else ev.cuda_time_total
for (name, shp, st), (n, us) in sorted(rows.items(), key=lambda kv: -kv[1][1]):
    print(f"{n:4d} x {name:16s} {us:9.1f} us  {shp}  {st}")
print("--- device kernels not from the library")
agg = {}
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CUDA and not ev.name.startswith("fz::") and "fz" not in ev.name[:6]:
        a = agg.setdefault(ev.name[:90], [0, 0.0]); a[0] += 1; a[1] += ev.device_time_total
for k, (n, us) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:20]:
    print(f"{n:4d} x {us:9.1f} us  {k}")
print("--- runtime copy calls and the operators around them")
evs = list(prof.events())
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
rt = [e for e in cpu if "emcpy" in e.name or "emset" in e.name]
chains = {}
for r in rt:
    enc = [e for e in cpu if e is not r and e.thread == r.thread and e.time_range.start <= r.time_range.start and e.time_range.end >= r.time_range.end]
    enc.sort(key=lambda e: e.time_range.start)
    key = (r.name, " > ".join(e.name for e in enc[-4:]), str(enc[-1].input_shapes)[:70] if enc else "")
    chains[key] = chains.get(key, 0) + 1
for k, n in sorted(chains.items(), key=lambda kv: -kv[1]):
    print(n, k)
agg = {}
for ev in evs:
    if ev.device_type == torch.autograd.DeviceType.CUDA and ("emcpy" in ev.name or "emset" in ev.name or "rocclr" in ev.name):
        a = agg.setdefault(ev.name[:90], [0, 0.0]); a[0] += 1; a[1] += ev.device_time_total
print(agg)
