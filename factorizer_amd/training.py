"""Training-step pieces either side of the network: the optimizer, the learning
rate schedule and checkpoint interchange of the BraTS bundle
(model_zoo/factorizer_brats23/configs/train.yaml:67-83, scripts/utils.py:10-31).

* `FlatAdamW` — torch.optim.AdamW semantics over ONE flat parameter / gradient / moment buffer;
  on device the update is one kernel (csrc/optim.hip, `fz_adamw_step`).
* `WarmupCosineSchedule` — MONAI's `monai.optimizers.WarmupCosineSchedule` (third-party, not under the
  reference tree; published formula restated): linear warm-up from `warmup_multiplier` to 1 over
  `warmup_steps`, then cosine decay to 0 at `t_total` (`cycles` = 0.5).
* `load_checkpoint` — what the bundle's `load_checkpoint` does through ignite's
  `Checkpoint.load_objects`: `objects[key].load_state_dict(checkpoint[key])`; the module
  `state_dict` keys are the reference's (tests/golden/g6_readme_model_keys.npz).
"""
from __future__ import annotations

import math
from typing import Any, Dict, Iterable

import torch

from . import _native as N


class FlatAdamW:
    """AdamW (decoupled weight decay, no amsgrad) over the trainable parameters of `module`, moved
    into one flat buffer (each `p.data` becomes a view, so `state_dict`/`load_state_dict` and the
    forward are unaffected).  `step()` packs the gradients (one multi-tensor copy, or none when
    `grad_views` — e.g. FlatGradSync.views after `finish()` — already alias `flat_grad`)."""

    def __init__(self, module_or_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat_grad=None,
                 grad_views=None):
        params = module_or_params.parameters() if isinstance(module_or_params, torch.nn.Module) else module_or_params
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32 or any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("FlatAdamW: fp32 parameters on one device")
        order = self.params if grad_views is None else list(grad_views.keys())
        if grad_views is not None and set(order) != set(self.params):
            raise ValueError("FlatAdamW: grad_views must cover exactly the trainable parameters")
        total = sum(p.numel() for p in order)
        self.flat_param = torch.empty(total, device=dev, dtype=dt)
        self.flat_grad = flat_grad if flat_grad is not None else torch.zeros(total, device=dev, dtype=dt)
        if self.flat_grad.numel() != total:
            raise ValueError("FlatAdamW: flat_grad size mismatch")
        self.exp_avg = torch.zeros(total, device=dev, dtype=dt)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=dt)
        self.grad_views, off = {}, 0
        with torch.no_grad():
            for p in order:  # same order as the gradient buffer
                n = p.numel()
                view = self.flat_param[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.grad_views[p] = grad_views[p] if grad_views is not None else self.flat_grad[off:off + n].view_as(p)
                off += n
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), tuple(betas), float(eps), float(weight_decay)
        self.base_lr = float(lr)
        self.t = 0

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        from . import pointwise as _PW
        _PW.join_wgrad_streams()  # no-op unless late_wgrad_join is on (pointwise._LateJoin)
        src = [p for p in self.params if p.grad is not None and p.grad.data_ptr() != self.grad_views[p].data_ptr()]
        if src:
            torch._foreach_copy_([self.grad_views[p] for p in src], [p.grad for p in src])
        for p in self.params:
            if p.grad is None:
                self.grad_views[p].zero_()
        self.t += 1
        b1, b2 = self.betas
        if self.flat_param.is_cuda:
            with torch.cuda.device(self.flat_param.device):
                rc = N.lib().fz_adamw_step(self.flat_param.data_ptr(), self.flat_grad.data_ptr(),
                                           self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                           self.flat_param.numel(), self.lr, b1, b2, self.eps, self.weight_decay,
                                           self.t, float(grad_scale), N.stream_ptr(self.flat_param))
            N.check(rc, "fz_adamw_step")
            return
        g = self.flat_grad * grad_scale
        self.flat_param.mul_(1.0 - self.lr * self.weight_decay)
        self.exp_avg.mul_(b1).add_(g, alpha=1.0 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1 ** self.t, 1.0 - b2 ** self.t
        denom = (self.exp_avg_sq.sqrt() / math.sqrt(bc2)).add_(self.eps)
        self.flat_param.addcdiv_(self.exp_avg, denom, value=-self.lr / bc1)

    def state_dict(self) -> Dict[str, Any]:
        return {"t": self.t, "lr": self.lr, "exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq}

    def load_state_dict(self, sd: Dict[str, Any]):
        self.t, self.lr = int(sd["t"]), float(sd["lr"])
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])


class WarmupCosineSchedule:
    """lr(step) = base_lr · λ(step);  λ = warmup_multiplier + (1 − warmup_multiplier)·step/warmup_steps for
    step < warmup_steps, else max(0, ½(1 + cos(π·2·cycles·progress))), progress = (step − warmup_steps) /
    max(1, t_total − warmup_steps).  The bundle: warmup_steps = num_epochs // 100, t_total = num_epochs + 1,
    warmup_multiplier = 0.1, stepped once per epoch (train.yaml:26-32, 79-83)."""

    def __init__(self, optimizer, warmup_steps: int, t_total: int, cycles: float = 0.5, last_epoch: int = -1,
                 warmup_multiplier: float = 0.0):
        self.optimizer, self.warmup_steps, self.t_total = optimizer, int(warmup_steps), int(t_total)
        self.cycles, self.warmup_multiplier = float(cycles), float(warmup_multiplier)
        self.base_lr = getattr(optimizer, "base_lr", None)
        if self.base_lr is None:
            self.base_lr = [g["lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self.step()

    def factor(self, step: int) -> float:
        if step < self.warmup_steps:
            f = float(step) / float(max(1.0, self.warmup_steps))
            return self.warmup_multiplier + (1 - self.warmup_multiplier) * f
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.cycles) * 2.0 * progress)))

    def step(self):
        self.last_epoch += 1
        f = self.factor(self.last_epoch)
        if isinstance(self.base_lr, list):
            for g, b in zip(self.optimizer.param_groups, self.base_lr):
                g["lr"] = b * f
        else:
            self.optimizer.lr = self.base_lr * f

    def get_last_lr(self):
        if isinstance(self.base_lr, list):
            return [g["lr"] for g in self.optimizer.param_groups]
        return [self.optimizer.lr]


def load_checkpoint(objects: Dict[str, Any], path_or_data, strict: bool = True, **torch_load_kw) -> Dict[str, Any]:
    """`objects[key].load_state_dict(checkpoint[key])` for every key (ignite `Checkpoint.load_objects`, as
    used by the bundle's scripts/utils.py:10-26); a bare `state_dict` is accepted for a single object."""
    data = torch.load(path_or_data, **torch_load_kw) if isinstance(path_or_data, (str, bytes)) or hasattr(path_or_data, "read") \
        else path_or_data
    if len(objects) == 1 and not any(k in data for k in objects) and all(torch.is_tensor(v) for v in data.values()):
        data = {next(iter(objects)): data}
    for key, obj in objects.items():
        if key not in data:
            raise KeyError(f"load_checkpoint: checkpoint has no entry {key!r}")
        if isinstance(obj, torch.nn.Module):
            obj.load_state_dict(data[key], strict=strict)
        else:
            obj.load_state_dict(data[key])
    return objects
