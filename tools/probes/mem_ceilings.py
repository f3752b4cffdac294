"""What this box's memory system gives a plain streaming launch: read-only, write-only, copy (framework kernels, 1 GiB)."""
import json, torch
n = 1 << 28
a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
def t(fn, it=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(it): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / it
nb = n * 4
out = {"bytes": nb,
       "read_only_sum_GBps": round(nb / t(lambda: a.sum()) / 1e6, 1),
       "read_only_amax_GBps": round(nb / t(lambda: a.amax()) / 1e6, 1),
       "write_only_fill_GBps": round(nb / t(lambda: b.fill_(1.0)) / 1e6, 1),
       "copy_read_plus_write_GBps": round(2 * nb / t(lambda: b.copy_(a)) / 1e6, 1),
       "scale_inplace_read_plus_write_GBps": round(2 * nb / t(lambda: a.mul_(1.0001)) / 1e6, 1)}
print(json.dumps(out))
