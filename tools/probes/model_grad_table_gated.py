"""tests/test_gpu_model.py::test_five_stage_model_vs_oracle_every_gradient as a table: device vs float64 oracle WITH THE DEVICE'S
ReLU GATES, next to fp32 oracle vs the same float64 run.  usage: model_grad_table_gated.py S patch B"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import torch
import test_gpu_model as T
from oracle import cpu_ref as O

S, patch, B = (int(sys.argv[1]),) * 3, int(sys.argv[2]), int(sys.argv[3])
torch.manual_seed(3)
model = T._model(S, patch)
sd = {k: v.clone() for k, v in model.state_dict().items()}
cfg = dict(widths=T.WIDTHS, strides=T.STRIDES, reshape=dict(head_dim=8, patch_size=patch), num_iters=5, solver="hals")
x = torch.rand(B, 4, *S); gy = torch.randn(B, 3, *S)
model = model.cuda()
yd, gates = T._device_relu_gates(model, x.cuda())
yd.backward(gy.cuda())


def run(dt, gated):
    it = iter(gates)
    def fm(xx, sdd, prefix, c):
        C, spatial = xx.shape[1], tuple(xx.shape[2:])
        z = O.linear_cf(xx, sdd[prefix + "in_proj.linear.weight"])
        gate = next(it)
        t = torch.relu(z).detach() + gate.to(z.dtype) * (z - z.detach()) if gated else torch.relu(z)
        m = O.nmf_forward(O.swm_forward(t, **c["reshape"]), sdd[prefix + "factorize.init.u0"], sdd[prefix + "factorize.init.v0"], 5, "hals", None)
        return O.linear_cf(O.swm_inverse(m, C, spatial, **c["reshape"]), sdd[prefix + "out_proj.linear.weight"], sdd[prefix + "out_proj.linear.bias"])
    prm = {k: v.clone().to(dt).requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and not k.endswith(("u0", "v0"))}
    full = {k: (v.to(dt) if v.is_floating_point() else v) for k, v in sd.items()}; full.update(prm)
    orig = O.fact_mixer; O.fact_mixer = fm
    try:
        y = O.factorizer_forward(x.to(dt), full, cfg)
        g = dict(zip(prm.keys(), torch.autograd.grad(y, list(prm.values()), gy.to(dt))))
    finally:
        O.fact_mixer = orig
    return g


g64 = run(torch.float64, True); g32 = run(torch.float32, True)
rows = []
for n, p in model.named_parameters():
    sc = g64[n].abs().max().item() + 1e-30
    rows.append((n, (p.grad.double().cpu() - g64[n]).abs().max().item() / sc, (g32[n].double() - g64[n]).abs().max().item() / sc))
rows.sort(key=lambda r: -r[1])
print("%-70s %9s %9s   (both against float64 with the device's gates)" % ("tensor", "dev max", "f32 max"))
for r in rows[:14]: print("%-70s %9.2e %9.2e" % r)
