"""How long does the host need to ENQUEUE one training step (no synchronisation)?"""
import sys, time, torch
sys.path.insert(0, '.')
import bench
import factorizer_amd as ft
from factorizer_amd.parallel import FlatGradSync
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ft.Factorizer(**bench.MODEL_KW).to(dev).train()
sync = FlatGradSync(model, num_buckets=2, overlap=True)
opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-5, fused=True)
x = torch.rand(2, 4, 128, 128, 128, device=dev)
t = (torch.rand(2, 3, 128, 128, 128, device=dev) > 0.5).float()
def step():
    sync.zero_grad()
    loss = ft.dice_bce_loss(model(x), t)
    loss.backward()
    sync.finish()
    opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter(); step(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.2f} ms, total {1e3*(t2-t0):.2f} ms")
# forward / backward split of the host time
t0 = time.perf_counter(); sync.zero_grad(); y = model(x); loss = ft.dice_bce_loss(y, t); t1 = time.perf_counter(); loss.backward(); t2 = time.perf_counter(); sync.finish(); opt.step(); t3 = time.perf_counter()
torch.cuda.synchronize()
print(f"host: fwd {1e3*(t1-t0):.2f} bwd {1e3*(t2-t1):.2f} opt {1e3*(t3-t2):.2f}")
