"""Soak: 300 training steps of the bench workload; memory must be flat and the loss finite."""
import sys, time, torch
sys.path.insert(0, '.')
import bench
import factorizer_amd as ft
from factorizer_amd import pointwise as PW
dev = torch.device('cuda', 0)
torch.manual_seed(0)
model = ft.Factorizer(**bench.MODEL_KW).to(dev).train()
sync = ft.FlatGradSync(model, num_buckets=2, overlap=True, late_wgrad_join=True)
opt = ft.FlatAdamW(model, lr=1e-4, weight_decay=1e-5, flat_grad=sync.flat, grad_views=sync.views)
x = torch.rand(2, 4, 128, 128, 128, device=dev)
t = (torch.rand(2, 3, 128, 128, 128, device=dev) > 0.5).float()
marks = {}
t0 = time.perf_counter()
for i in range(300):
    sync.zero_grad()
    loss = ft.dice_bce_loss(model(x), t)
    loss.backward()
    sync.finish()
    opt.step()
    if i in (20, 150, 299):
        torch.cuda.synchronize()
        marks[i] = (round(torch.cuda.memory_allocated() / 2**30, 3), round(torch.cuda.max_memory_allocated() / 2**30, 3),
                    round(torch.cuda.memory_reserved() / 2**30, 3), round(loss.item(), 5), len(PW._LateJoin.keep), len(PW._LateJoin.owed))
torch.cuda.synchronize()
print("steps/s", 300 / (time.perf_counter() - t0), marks)
assert marks[299][0] <= marks[20][0] + 0.01 and marks[299][2] <= marks[20][2] + 0.5, "memory grows"
assert all(m[3] == m[3] for m in marks.values())
print("soak ok")
