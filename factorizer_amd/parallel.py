"""Data-parallel training over the GPUs of one node: one process per GPU, gradients averaged
with ONE flat all-reduce per bucket over RCCL/xGMI (``torch.distributed`` backend "nccl" is
RCCL on ROCm; "gloo" on CPU for the tests).

Replaces the reference's only multi-GPU mechanism, the MONAI-bundle
``DistributedDataParallel`` wrapper (model_zoo/factorizer_brats23/configs/
train_multigpu.yaml:3-6,27).  The Factorizer has 5.86 M parameters (23.4 MB fp32): the
payload is latency-bound, so gradients live in a single flat buffer (p.grad are views into
it — no gather/scatter copies) and each bucket is reduced by one collective, launched from
autograd hooks as soon as the bucket's last gradient is produced so that it overlaps the rest
of the backward pass.  The weight-gradient kernels write straight into the bucket memory (gradbuf.py): only the
small vectors (biases, LayerNorm parameters) are packed by a multi-tensor copy before a bucket's collective.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class FlatGradSync:
    def __init__(self, module: torch.nn.Module, process_group=None, num_buckets: int = 4,
                 overlap: bool = True, late_wgrad_join: bool = False, force_collectives: bool = False,
                 deferred_finishes: bool = True):
        # force_collectives: run the hook-launched all-reduces and finish() even in a one-rank group
        # (bench.py --force-dist: the N = 1 line then executes the same RCCL path as N > 1)
        # late_wgrad_join: the side-stream weight-gradient kernels of the deep blocks are awaited once,
        # right before the gradients are read (bucket collectives / optimizer), not at the end of every
        # block backward (pointwise._LateJoin)
        self.late_join = bool(late_wgrad_join)
        if self.late_join:
            from . import pointwise as _PW
            _PW.late_wgrad_join(True)
        if deferred_finishes:   # the finish reductions of the weight-gradient launches: queued, run in front of each bucket's
            from . import pointwise as _PW   # collective and at the end of the backward (pointwise._Defer)
            _PW.defer_finishes(True)
        self.module = module
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self._param_ids = frozenset(id(p) for p in self.params)   # deferred finishes are armed for these parameters only
        dev, dt = self.params[0].device, self.params[0].dtype
        from .training import flat_align   # same 16-byte-aligned layout as FlatAdamW's buffers (it aliases this one)
        total = sum(flat_align(p.numel()) for p in self.params)
        self.flat = torch.zeros(total, device=dev, dtype=dt)
        # autograd produces gradients roughly in reverse registration order: bucket 0 = the
        # parameters registered LAST (decoder / head), reduced first.
        order = list(reversed(self.params))
        per = (total + num_buckets - 1) // max(num_buckets, 1)
        self.buckets, off, cur, cur_n = [], 0, [], 0
        start = 0
        self.views = {}
        for p in order:
            n = p.numel()
            self.views[p] = self.flat[off:off + n].view_as(p)
            cur.append(p)
            off += flat_align(n)
            cur_n += flat_align(n)
            if cur_n >= per:
                self.buckets.append({"params": cur, "lo": start, "hi": off})
                cur, cur_n, start = [], 0, off
        if cur:
            self.buckets.append({"params": cur, "lo": start, "hi": off})
        # the weight-gradient launches write straight into the bucket memory (gradbuf.py; a FlatAdamW built on these
        # views re-registers them once it has moved the parameters into its own flat buffer)
        from . import gradbuf as _GB
        _GB.register(self.flat, self.views)
        self._pending = []
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        self.active = self.world > 1 or (force_collectives and dist.is_initialized())
        self.overlap = overlap and self.active
        if self.overlap:
            for bi, b in enumerate(self.buckets):
                for p in b["params"]:
                    p.register_post_accumulate_grad_hook(self._make_hook(bi))

    # ---- parameters / buffers start identical on every rank (incl. NMF init.u0/v0) ----
    def broadcast_state(self, src: int = 0):
        if self.world == 1:
            return
        for t in list(self.module.parameters()) + list(self.module.buffers()):
            dist.broadcast(t.data, src, group=self.group)

    def _make_hook(self, bi):
        def hook(_p):
            self._ready[bi] += 1
            if self._ready[bi] == len(self.buckets[bi]["params"]):
                self._launch(bi)
        return hook

    def _launch(self, bi):
        b = self.buckets[bi]
        self._launched[bi] = True
        from . import pointwise as _PW
        if self.late_join:
            _PW.wait_wgrad_streams()  # (ownership is checked in finish(), once every gradient is assigned)
        _PW.flush_finishes()          # the bucket's gradients must be complete before they are packed and reduced
        # pack the bucket's gradients into the flat buffer with one multi-tensor copy (autograd
        # produced them as separate tensors: assigning, not accumulating, costs no kernel)
        ps = [p for p in b["params"] if p.grad is not None and p.grad.data_ptr() != self.views[p].data_ptr()]
        if ps:   # (weight gradients are already in place — gradbuf.py; biases and LayerNorm parameters are packed here)
            torch._foreach_copy_([self.views[p] for p in ps], [p.grad for p in ps])
        missing = [p for p in b["params"] if p.grad is None]
        for p in missing:
            self.views[p].zero_()
        work = dist.all_reduce(self.flat[b["lo"]:b["hi"]], op=dist.ReduceOp.SUM, group=self.group,
                               async_op=True)
        self._pending.append(work)

    def zero_grad(self):
        """Gradients are dropped (set to None), so the backward pass ASSIGNS fresh tensors instead of
        launching one accumulate kernel per parameter."""
        from . import pointwise as _PW
        _PW.reset_late_join()
        for p in self.params:
            p.grad = None
        from . import gradbuf as _GB
        _GB.release(self.flat)
        _PW.arm_deferred_finishes(self._param_ids)
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)

    def finish(self, average: bool = True) -> float:
        """Call after backward(): launches the buckets whose hooks did not all fire (a parameter that took
        no part in the loss never fires its hook: its slice of the flat buffer is zero-filled and reduced
        like the rest, as DistributedDataParallel(find_unused_parameters=True) does), waits for the
        collectives and averages.  The set of unused parameters must be the SAME on every rank (as for a static graph
        under DDP): a bucket is launched from a hook on the ranks where all its parameters were used and from here on the
        others, and collectives must be issued in one order everywhere.  A parameter unused on EVERY rank still receives
        a zero gradient here (weight decay and a step count under FlatAdamW at N > 1, where a single rank would skip
        it): freeze such parameters (`requires_grad_(False)`) instead of leaving them unused.
        `average=False` leaves the SUM in the buffer and returns the factor
        1/world for the optimizer to apply (`FlatAdamW.step(grad_scale=...)`: no extra pass)."""
        from . import pointwise as _PW
        if self.late_join:
            _PW.join_wgrad_streams()
        _PW.flush_finishes(disarm=True)
        if not self.active:
            return 1.0
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:
                self._launch(bi)
        for w in self._pending:
            w.wait()
        self._pending = []
        scale = 1.0 / self.world
        if average and self.world > 1:
            self.flat.div_(self.world)
            scale = 1.0
        for p in self.params:  # the optimizer reads the reduced gradients through the flat views
            p.grad = self.views[p]
        self._ready = [0] * len(self.buckets)
        self._launched = [False] * len(self.buckets)
        return scale
