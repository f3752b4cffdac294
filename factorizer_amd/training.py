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
from . import gradbuf as _GB

FLAT_ALIGN = 4  # floats: every parameter starts on a 16-byte boundary of the flat buffers (float4 kernels, 32-byte weight reads)


def flat_align(n: int) -> int:
    return (n + FLAT_ALIGN - 1) // FLAT_ALIGN * FLAT_ALIGN


class FlatAdamW:
    """AdamW (decoupled weight decay, no amsgrad) over the trainable parameters of `module`, moved
    into one flat buffer (each `p.data` becomes a view, so `state_dict`/`load_state_dict` and the
    forward are unaffected).  `step()` packs the gradients (one multi-tensor copy, or none when
    `grad_views` — e.g. FlatGradSync.views after `finish()` — already alias `flat_grad`).

    Semantics are torch.optim.AdamW's (train.yaml:72-76), including: a parameter whose `.grad` is None
    is skipped entirely (no decay, no moment update, its step count does not advance), and
    `state_dict()` / `load_state_dict()` use torch's layout — `{"state": {i: {"step", "exp_avg",
    "exp_avg_sq"}}, "param_groups": [...]}` with i the position in the parameter list — so optimizer
    checkpoints interchange with the reference recipe's (train.yaml:354-358)."""

    def __init__(self, module_or_params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, flat_grad=None,
                 grad_views=None, deferred_finishes: bool = True):
        # deferred_finishes: this optimizer owns the step (gradients are None from zero_grad() to the backward, nothing reads
        # them before step()), so the tiny fixed-order reductions that end every weight-gradient launch may be queued during
        # the backward and run as one grid at its end (pointwise._Defer; csrc/finish.h) — same sums, same order
        if deferred_finishes:
            from . import pointwise as _PW
            _PW.defer_finishes(True)
        params = module_or_params.parameters() if isinstance(module_or_params, torch.nn.Module) else module_or_params
        self.all_params = list(params)               # torch's param_groups[0]["params"] order
        self.params = [p for p in self.all_params if p.requires_grad]
        if not self.params:
            raise ValueError("FlatAdamW: no trainable parameters")
        self._param_ids = frozenset(id(p) for p in self.params)   # deferred finishes are armed for these parameters only
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32 or any(p.dtype != dt or p.device != dev for p in self.params):
            raise ValueError("FlatAdamW: fp32 parameters on one device")
        order = self.params if grad_views is None else list(grad_views.keys())
        if grad_views is not None and set(order) != set(self.params):
            raise ValueError("FlatAdamW: grad_views must cover exactly the trainable parameters")
        # every parameter starts 16-byte aligned (padding elements stay zero in all four buffers: a zero parameter with a
        # zero gradient is a fixed point of the update), so any sub-range launch of the float4 kernel is aligned too
        total = sum(flat_align(p.numel()) for p in order)
        self.flat_param = torch.zeros(total, device=dev, dtype=dt)
        self.flat_grad = flat_grad if flat_grad is not None else torch.zeros(total, device=dev, dtype=dt)
        if self.flat_grad.numel() != total:
            raise ValueError("FlatAdamW: flat_grad size mismatch")
        self.exp_avg = torch.zeros(total, device=dev, dtype=dt)
        self.exp_avg_sq = torch.zeros(total, device=dev, dtype=dt)
        self.grad_views, self.offsets, off = {}, {}, 0
        with torch.no_grad():
            for p in order:  # same order as the gradient buffer
                n = p.numel()
                view = self.flat_param[off:off + n].view_as(p)
                view.copy_(p.data)
                p.data = view
                self.grad_views[p] = grad_views[p] if grad_views is not None else self.flat_grad[off:off + n].view_as(p)
                self.offsets[p] = off
                off += flat_align(n)
        self.order = order
        # the weight-gradient launches write straight into flat_grad (gradbuf.py): no packing copy in step()
        _GB.register(self.flat_grad, self.grad_views)
        # ONE persistent torch-style group: schedulers and utilities written against torch optimizers read AND write
        # it (g["lr"] = ...); lr / betas / eps / weight_decay / base_lr of this object are views of it
        self._group = {"lr": float(lr), "initial_lr": float(lr), "betas": tuple(betas), "eps": float(eps),
                       "weight_decay": float(weight_decay), "amsgrad": False, "params": self.all_params}
        self.steps = {p: 0 for p in self.params}     # per-parameter step counts, as torch keeps them

    @property
    def t(self) -> int:
        return max(self.steps.values())

    def __deepcopy__(self, memo):
        """nn.Parameter deep-copies by cloning, which would untie the copies from the copied flat buffer:
        re-point them (scripts/utils.py:21 deep-copies {network, optimizer} before loading)."""
        import copy
        new = object.__new__(type(self))
        memo[id(self)] = new
        for k, v in self.__dict__.items():
            setattr(new, k, copy.deepcopy(v, memo))
        with torch.no_grad():
            for q in new.order:
                lo, n = new.offsets[q], q.numel()
                q.data = new.flat_param[lo:lo + n].view_as(q)
        _GB.register(new.flat_grad, new.grad_views)   # the copy's weight gradients land in the copy's buffer
        return new

    @property
    def param_groups(self):
        """torch-style groups: one persistent dict (writes to it — a scheduler's g["lr"] = ... — take effect)."""
        return [self._group]

    lr = property(lambda self: float(self._group["lr"]), lambda self, v: self._group.__setitem__("lr", float(v)))
    base_lr = property(lambda self: float(self._group["initial_lr"]), lambda self, v: self._group.__setitem__("initial_lr", float(v)))
    betas = property(lambda self: tuple(self._group["betas"]), lambda self, v: self._group.__setitem__("betas", tuple(v)))
    eps = property(lambda self: float(self._group["eps"]), lambda self, v: self._group.__setitem__("eps", float(v)))
    weight_decay = property(lambda self: float(self._group["weight_decay"]),
                            lambda self, v: self._group.__setitem__("weight_decay", float(v)))

    def zero_grad(self, set_to_none: bool = True):
        from . import pointwise as _PW
        _PW.reset_late_join()
        for p in self.params:
            p.grad = None
        _GB.release(self.flat_grad)
        _PW.arm_deferred_finishes(self._param_ids)

    def _update(self, lo, hi, t, grad_scale):
        b1, b2 = self.betas
        P, G, M, V = (x[lo:hi] for x in (self.flat_param, self.flat_grad, self.exp_avg, self.exp_avg_sq))
        if self.flat_param.is_cuda:
            with torch.cuda.device(self.flat_param.device):
                rc = N.lib().fz_adamw_step(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), hi - lo, self.lr,
                                           b1, b2, self.eps, self.weight_decay, t, float(grad_scale),
                                           N.stream_ptr(self.flat_param))
            N.check(rc, "fz_adamw_step")
            return
        g = G * grad_scale
        P.mul_(1.0 - self.lr * self.weight_decay)
        M.mul_(b1).add_(g, alpha=1.0 - b1)
        V.mul_(b2).addcmul_(g, g, value=1.0 - b2)
        bc1, bc2 = 1.0 - b1 ** t, 1.0 - b2 ** t
        denom = (V.sqrt() / math.sqrt(bc2)).add_(self.eps)
        P.addcdiv_(M, denom, value=-self.lr / bc1)

    @torch.no_grad()
    def step(self, grad_scale: float = 1.0):
        from . import pointwise as _PW
        _PW.join_wgrad_streams()  # no-op unless late_wgrad_join is on (pointwise._LateJoin)
        _PW.flush_finishes(disarm=True)   # (the end of the backward already did: nothing is queued unless backward() raised)
        src = [p for p in self.params if p.grad is not None and p.grad.data_ptr() != self.grad_views[p].data_ptr()]
        if src:
            torch._foreach_copy_([self.grad_views[p] for p in src], [p.grad for p in src])
        # one launch per maximal run of buffer-adjacent parameters that have a gradient and share a step
        # count: ONE launch over the whole buffer in the usual case
        run = None  # [lo, hi, t]
        for p in self.order:
            if p.grad is None:
                if run:
                    self._update(*run, grad_scale)
                    run = None
                continue
            self.steps[p] += 1
            lo = self.offsets[p]
            hi = lo + flat_align(p.numel())   # (the padding rides along: zeros stay zeros)
            if run and run[1] == lo and run[2] == self.steps[p]:
                run[1] = hi
            else:
                if run:
                    self._update(*run, grad_scale)
                run = [lo, hi, self.steps[p]]
        if run:
            self._update(*run, grad_scale)

    def state_dict(self) -> Dict[str, Any]:
        state = {}
        for i, p in enumerate(self.all_params):
            if p in self.steps and self.steps[p] > 0:
                lo, n = self.offsets[p], p.numel()
                state[i] = {"step": torch.tensor(float(self.steps[p])),
                            "exp_avg": self.exp_avg[lo:lo + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[lo:lo + n].view_as(p).clone()}
        group = {"lr": self.lr, "initial_lr": self.base_lr, "betas": self.betas, "eps": self.eps,
                 "weight_decay": self.weight_decay, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.all_params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd: Dict[str, Any]):
        """Accepts torch.optim.AdamW's state_dict (the recipe's checkpoints) — and this class's."""
        if "param_groups" not in sd:  # round-1 flat layout
            for p in self.params:
                self.steps[p] = int(sd["t"])
            self.lr = float(sd["lr"])
            src = 0  # that layout packed the parameters without alignment padding
            with torch.no_grad():
                for p in self.order:
                    lo, n = self.offsets[p], p.numel()
                    self.exp_avg[lo:lo + n].copy_(sd["exp_avg"][src:src + n])
                    self.exp_avg_sq[lo:lo + n].copy_(sd["exp_avg_sq"][src:src + n])
                    src += n
            return
        groups = sd["param_groups"]
        ids = [i for g in groups for i in g["params"]]
        if len(ids) != len(self.all_params):
            raise ValueError(f"FlatAdamW.load_state_dict: checkpoint has {len(ids)} parameters, optimizer has "
                             f"{len(self.all_params)}")
        g0 = groups[0]
        for g in groups[1:]:   # one fused launch = one set of hyperparameters
            for k in ("lr", "betas", "eps", "weight_decay"):
                if k in g and k in g0 and (tuple(g[k]) if k == "betas" else float(g[k])) != (tuple(g0[k]) if k == "betas" else float(g0[k])):
                    raise ValueError(f"FlatAdamW.load_state_dict: param_groups differ in {k!r}: per-group hyperparameters "
                                     "are not supported")
        self.lr = float(g0["lr"])
        self.base_lr = float(g0.get("initial_lr", self.base_lr))
        self.betas, self.eps = tuple(g0.get("betas", self.betas)), float(g0.get("eps", self.eps))
        self.weight_decay = float(g0.get("weight_decay", self.weight_decay))
        if g0.get("amsgrad", False):
            raise ValueError("FlatAdamW: amsgrad checkpoints are not supported")
        with torch.no_grad():
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for p in self.params:
                self.steps[p] = 0
            for pos, key in enumerate(ids):
                st = sd["state"].get(key)
                p = self.all_params[pos]
                if st is None or p not in self.steps:
                    continue
                lo, n = self.offsets[p], p.numel()
                self.steps[p] = int(float(st["step"]))
                self.exp_avg[lo:lo + n].view_as(p).copy_(st["exp_avg"])
                self.exp_avg_sq[lo:lo + n].view_as(p).copy_(st["exp_avg_sq"])


class WarmupCosineSchedule:
    """lr(step) = base_lr · λ(step);  λ = warmup_multiplier + (1 − warmup_multiplier)·step/warmup_steps for
    step < warmup_steps, else max(0, ½(1 + cos(π·2·cycles·progress))), progress = (step − warmup_steps) /
    max(1, t_total − warmup_steps).  The bundle: warmup_steps = num_epochs // 100, t_total = num_epochs + 1,
    warmup_multiplier = 0.1, stepped once per epoch (train.yaml:26-32, 79-83).  MONAI's class is a
    torch `LambdaLR`; `state_dict()` / `load_state_dict()` carry that layout's `last_epoch` / `base_lrs`."""

    def __init__(self, optimizer, warmup_steps: int, t_total: int, cycles: float = 0.5, last_epoch: int = -1,
                 warmup_multiplier: float = 0.0):
        self.optimizer, self.warmup_steps, self.t_total = optimizer, int(warmup_steps), int(t_total)
        self.cycles, self.warmup_multiplier = float(cycles), float(warmup_multiplier)
        self._flat = isinstance(optimizer, FlatAdamW)
        if self._flat:
            self.base_lrs = [optimizer.base_lr]
        else:
            for g in optimizer.param_groups:
                g.setdefault("initial_lr", g["lr"])
            self.base_lrs = [g["initial_lr"] for g in optimizer.param_groups]
        self.last_epoch = last_epoch
        self._step_count = 0
        self.step()

    @property
    def base_lr(self):
        return self.base_lrs[0] if self._flat else list(self.base_lrs)

    def factor(self, step: int) -> float:
        if step < self.warmup_steps:
            f = float(step) / float(max(1.0, self.warmup_steps))
            return self.warmup_multiplier + (1 - self.warmup_multiplier) * f
        progress = float(step - self.warmup_steps) / float(max(1, self.t_total - self.warmup_steps))
        return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(self.cycles) * 2.0 * progress)))

    def _apply(self):
        f = self.factor(self.last_epoch)
        if self._flat:
            self.optimizer.lr = self.base_lrs[0] * f
        else:
            for g, b in zip(self.optimizer.param_groups, self.base_lrs):
                g["lr"] = b * f

    def step(self):
        self.last_epoch += 1
        self._step_count += 1
        self._apply()

    def get_last_lr(self):
        if self._flat:
            return [self.optimizer.lr]
        return [g["lr"] for g in self.optimizer.param_groups]

    def state_dict(self) -> Dict[str, Any]:
        return {"last_epoch": self.last_epoch, "base_lrs": list(self.base_lrs), "_step_count": self._step_count,
                "_last_lr": self.get_last_lr(), "warmup_steps": self.warmup_steps, "t_total": self.t_total,
                "cycles": self.cycles, "warmup_multiplier": self.warmup_multiplier}

    def load_state_dict(self, sd: Dict[str, Any]):
        self.last_epoch = int(sd["last_epoch"])
        self.base_lrs = [float(b) for b in sd.get("base_lrs", self.base_lrs)]
        self._step_count = int(sd.get("_step_count", self.last_epoch + 1))
        for k in ("warmup_steps", "t_total"):
            if k in sd:
                setattr(self, k, int(sd[k]))
        for k in ("cycles", "warmup_multiplier"):
            if k in sd:
                setattr(self, k, float(sd[k]))
        if self._flat:
            self.optimizer.base_lr = self.base_lrs[0]
        self._apply()


def load_checkpoint(objects: Dict[str, Any], path_or_data, strict: bool = True, inplace: bool = False,
                    **torch_load_kw) -> Dict[str, Any]:
    """The bundle's `load_checkpoint` (scripts/utils.py:10-26): deep-copy `objects`, then
    `objects[key].load_state_dict(checkpoint[key])` for every key (ignite `Checkpoint.load_objects`), and
    return the loaded COPIES — its `load_checkpoints` ensemble (inference.yaml:141-142) relies on every
    call yielding independent objects.  `inplace=True` loads into the objects passed in.  A bare
    `state_dict` is accepted for a single object."""
    data = torch.load(path_or_data, **torch_load_kw) if isinstance(path_or_data, (str, bytes)) or hasattr(path_or_data, "read") \
        else path_or_data
    if len(objects) == 1 and not any(k in data for k in objects) and all(torch.is_tensor(v) for v in data.values()):
        data = {next(iter(objects)): data}
    for key in objects:
        if key not in data:
            raise KeyError(f"load_checkpoint: checkpoint has no entry {key!r}")
    if not inplace:
        import copy
        objects = copy.deepcopy(objects)   # one memo: a model and the optimizer over its flat buffer stay tied
    for key, obj in objects.items():
        if isinstance(obj, torch.nn.Module):
            obj.load_state_dict(data[key], strict=strict)
        else:
            obj.load_state_dict(data[key])
    return objects


def load_checkpoints(objects: Dict[str, Any], paths: Iterable, **kw):
    """scripts/utils.py:29-32: one independent loaded copy of `objects` per checkpoint (fold ensemble)."""
    return [load_checkpoint(objects, p, **kw) for p in paths]
