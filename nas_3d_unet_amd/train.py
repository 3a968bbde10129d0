"""Train-step / search-step harness: the build's counterpart of the reference's hot loops
(train.py:117-128: zero_grad -> forward -> Dice -> backward -> Adam.step;
 search.py:211-238: architecture step on the validation batch, then weight step on the train batch).

MI355X-first mechanics (none of which the reference has):
  * all kernel weights live in ONE flat fp32 buffer, their gradients in a second one; backward
    kernels write gradients in place (no per-tensor accumulate launches) and ONE fused Adam
    launch updates everything (n3d_adam_step);
  * forward + backward (+ Adam) of a fixed-shape step is captured once into a HIP graph and
    replayed, which removes the per-kernel host launch cost (~300 launches per step);
  * data parallel: one process per GPU, the flat gradient buffer is all-reduced with RCCL
    (torch.distributed backend "nccl") in a few large buckets, issued on a side stream as soon as
    the backward graph segment that fills the bucket has been enqueued.
"""
from __future__ import annotations

import os
import types

import torch
import torch.distributed as dist

from . import fused as _fused
from . import kernels as K
from .loss import WeightedDiceLoss


def _loss_of(model, loss_fn, x, t):
    """Dice loss of the model on (x, t): through the model's fused head + loss path when it has one"""
    fl = getattr(model, "forward_loss", None)
    if fl is not None and isinstance(loss_fn, WeightedDiceLoss):
        return fl(x, t, loss_fn.smooth)[0]
    return loss_fn(model(x), t)


class FlatParams:
    """Re-homes a list of parameters into one flat buffer (+ a flat gradient buffer)."""

    def __init__(self, params, device):
        self.params = [p for p in params]
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + 3) // 4 * 4  # keep every tensor 16-byte aligned
        self.numel = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.grad = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.step = torch.zeros(1, dtype=torch.int32, device=device)
        self.offsets = offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = self.grad[o:o + p.numel()].view(p.shape)
                p._n3d_grad = g  # backward kernels write here (programs.py / kernels.grad_target)
                p.grad = g

    def adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, lr_dev=None):
        K.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step, lr, betas[0], betas[1], eps,
                    weight_decay, grad_scale, True, lr_dev)


class PlateauLR:
    """Host logic of torch.optim.lr_scheduler.ReduceLROnPlateau as the reference uses it
    (`ReduceLROnPlateau(optim, factor=0.5)`, train.py:50; search.py:105-106; stepped once per epoch with the validation
    loss, train.py:77; search.py:155-156): mode 'min', relative threshold 1e-4, patience 10, cooldown 0, min_lr 0,
    eps 1e-8.  `set_lr(new_lr)` is called when the rate changes; the trainers keep the rate in a device scalar that
    the Adam kernel reads, so a captured HIP graph follows the schedule without re-capture."""

    def __init__(self, get_lr, set_lr, factor=0.5, patience=10, threshold=1e-4, cooldown=0, min_lr=0.0, eps=1e-8):
        if factor >= 1.0:
            raise ValueError("Factor should be < 1.0.")
        self.get_lr, self.set_lr = get_lr, set_lr
        self.factor, self.patience, self.threshold, self.cooldown, self.min_lr, self.eps = factor, patience, threshold, cooldown, min_lr, eps
        self.best = float("inf")
        self.num_bad_epochs = 0
        self.cooldown_counter = 0
        self.last_epoch = 0

    def step(self, metric):
        current = float(metric)
        self.last_epoch += 1
        if current < self.best * (1.0 - self.threshold):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.cooldown_counter > 0:
            self.cooldown_counter -= 1
            self.num_bad_epochs = 0
        if self.num_bad_epochs > self.patience:
            old = self.get_lr()
            new = max(old * self.factor, self.min_lr)
            if old - new > self.eps:
                self.set_lr(new)
            self.cooldown_counter = self.cooldown
            self.num_bad_epochs = 0
        return self.get_lr()


class GradSync:
    """Bucketed mean all-reduce of one flat gradient buffer (works with RCCL on GPUs and gloo on CPU tensors).

    xGMI is point-to-point (7 links per GPU) and the whole payload is 2-7 MB (searched net) / 27 MB
    (supernet), so the collective is latency-bound: a few large buckets, never one call per tensor."""

    def __init__(self, flat_grad, process_group=None, n_buckets=2, comm_stream=None):
        self.g = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        nb = max(1, min(n_buckets, n // 4 if n >= 4 else 1))
        self.edges = [n * i // nb // 4 * 4 for i in range(nb)] + [n]
        self.comm_stream = comm_stream
        self.force = dist.is_initialized() and os.environ.get("N3D_FORCE_DP") == "1"  # see Trainer.dp_path

    def all_reduce(self):
        """sum-reduce every bucket; callers divide by world size (folded into the Adam kernel)."""
        if self.world == 1 and not self.force:
            return
        if self.comm_stream is not None:
            cs = self.comm_stream
            cs.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(cs):
                for a, b in zip(self.edges[:-1], self.edges[1:]):
                    dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.pg)
            torch.cuda.current_stream().wait_stream(cs)
        else:
            for a, b in zip(self.edges[:-1], self.edges[1:]):
                dist.all_reduce(self.g[a:b], op=dist.ReduceOp.SUM, group=self.pg)


def flatten_params(params, device=None):
    """(flat parameter tensor, flat gradient tensor): re-homes every parameter (and its .grad) as a view."""
    params = list(params)
    device = device if device is not None else params[0].device
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    grad = torch.zeros(n, dtype=torch.float32, device=device)
    with torch.no_grad():
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = grad[o:o + p.numel()].view(p.shape)
    return flat, grad, offs


class Trainer:
    """One searched-net (or any model built from nas_3d_unet_amd ops) training step.

    step(x, t): x (B,4,S,S,S), t (B,3,S,S,S) fp32 device tensors -> loss (0-d device tensor,
    no host sync).  With graph=True the first call captures, later calls replay."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph=True, process_group=None,
                 n_buckets=1, params=None):
        self.model = model
        self.loss_fn = WeightedDiceLoss()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(model.parameters()).device
        plist = list(params) if params is not None else list(model.parameters())
        self.fp = FlatParams(plist, self.device)
        self.use_graph = graph
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        # N3D_FORCE_DP=1: take the multi-GPU code path (eager all-reduce on the comm stream + eager Adam after the graph)
        # even in a 1-rank group -- lets a single-GPU box exercise exactly what N > 1 runs
        self.dp_path = self.world > 1 or (dist.is_initialized() and os.environ.get("N3D_FORCE_DP") == "1")
        self.n_buckets = max(1, n_buckets)
        self._graph = None
        self._static_x = self._static_t = self._static_loss = None
        # one bucket = nothing to overlap: the collective is issued from the step's own stream (a hop through a second
        # stream costs two cross-stream waits, ~0.3 ms per step around a graph launch); N3D_COMM_STREAM=1 restores the hop
        want_cs = self.dp_path and (self.n_buckets > 1 or os.environ.get("N3D_COMM_STREAM") == "1")
        self._comm_stream = torch.cuda.Stream(device=self.device) if want_cs else None
        self.ctx = K.StepContext(self.device)  # batched weight packing + deferred wgrad reductions
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)  # read by the Adam kernel
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.scheduler = PlateauLR(lambda: self.lr, self.set_lr)  # train.py:50: ReduceLROnPlateau(factor=0.5)
        self.sync = GradSync(self.fp.grad, self.pg, self.n_buckets, self._comm_stream)
        if self.world > 1:
            dist.broadcast(self.fp.flat, src=0, group=self.pg)

    # -- pieces ---------------------------------------------------------------------------------
    def _fwd_bwd(self, x, t):
        with K.step_context(self.ctx):
            self.ctx.pack_all()            # one launch packs every conv weight for this step
            loss = _loss_of(self.model, self.loss_fn, x, t)
            prev, _fused.REUSE_GRAD_OUTPUT = _fused.REUSE_GRAD_OUTPUT, True   # this backward is all ours (no hooks, no retain)
            try:
                loss.backward(self._one)  # seed gradient kept resident: no fill launch per step
            finally:
                _fused.REUSE_GRAD_OUTPUT = prev
            self.ctx.flush_final()         # one launch finishes every weight-gradient reduction
        if not self.ctx.frozen:
            self.ctx.freeze()              # first pass only recorded which weights / layouts are needed
        return loss.detach()

    def _allreduce(self):
        self.sync.all_reduce()

    def _update(self):
        self.fp.adam(self.lr, self.betas, self.eps, 0.0, 1.0 / self.world, self.lr_dev)

    def set_lr(self, lr):
        """new learning rate for the following steps (also inside an already captured graph)"""
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    # -- public ---------------------------------------------------------------------------------
    def step(self, x, t):
        if not self.use_graph:
            loss = self._fwd_bwd(x, t)
            if self.dp_path:
                self._allreduce()
            self._update()
            return loss
        if self._graph is None:
            self._capture(x, t)
        if x.shape != self._static_x.shape or t.shape != self._static_t.shape:
            # a batch of another shape (the reference's generator yields a smaller last batch of an epoch): the captured
            # graph is for one shape only, so this step runs eagerly (same kernels, same update)
            loss = self._fwd_bwd(x, t)
            if self.dp_path:
                self._allreduce()
            self._update()
            return loss
        self._static_x.copy_(x)
        self._static_t.copy_(t)
        self._graph.replay()
        if self.dp_path:
            self._allreduce()
            self._update()
        return self._static_loss

    def _capture(self, x, t):
        self._static_x = x.clone()
        self._static_t = t.clone()
        # warm-up on a side stream (allocator + lazy module state), restoring the weights afterwards
        keep = self.fp.flat.clone()
        s = torch.cuda.Stream(device=self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                self._fwd_bwd(self._static_x, self._static_t)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self.fp.flat.copy_(keep)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):  # RCCL's watchdog thread may touch the runtime meanwhile
            self._static_loss = self._fwd_bwd(self._static_x, self._static_t)
            if not self.dp_path:
                self._update()
        self._graph = g


class SearchTrainer:
    """Supernet search step, first-order DARTS as in the reference (search.py:211-238):
    architecture pass on the validation batch (Adam on the four alpha matrices), then weight pass on the
    training batch (Adam on the kernel weights); both Adam instances use torch defaults (search.py:103-104).

    Exact-equivalent savings over the reference: the architecture pass does not compute weight gradients
    (the reference computes and discards them, search.py:231) and the weight pass does not compute alpha
    gradients -- requires_grad is switched per pass, so the corresponding kernels are simply not launched."""

    def __init__(self, shell, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph=True):
        self.model = shell
        self.loss_fn = WeightedDiceLoss()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(shell.parameters()).device
        self.kparams = list(shell.kernel.parameters())
        self.aparams = list(shell.alphas())
        self.fp = FlatParams(self.kparams, self.device)
        self.aflat, self.agrad, aoffs = flatten_params(self.aparams, self.device)
        self.a_m = torch.zeros_like(self.aflat)
        self.a_v = torch.zeros_like(self.aflat)
        self.a_step = torch.zeros(1, dtype=torch.int32, device=self.device)
        # the alphas' Adam state in the shape checkpoint.adam_state_dict reads (optim_shell, search.py:103)
        self.afp = types.SimpleNamespace(params=self.aparams, offsets=aoffs, exp_avg=self.a_m, exp_avg_sq=self.a_v, step=self.a_step)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.ctx = K.StepContext(self.device)
        self.use_graph = graph
        self._graph = None
        # two learning rates (search.py:103-106): alphas ("shell") and kernel weights, each on its own plateau schedule
        self.lr_shell, self.lr_kernel = float(lr), float(lr)
        self.lr_shell_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
        self.lr_kernel_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
        self.shell_scheduler = PlateauLR(lambda: self.lr_shell, self.set_shell_lr)
        self.kernel_scheduler = PlateauLR(lambda: self.lr_kernel, self.set_kernel_lr)

    def set_shell_lr(self, lr):
        self.lr_shell = float(lr)
        self.lr_shell_dev.fill_(self.lr_shell)

    def set_kernel_lr(self, lr):
        self.lr_kernel = float(lr)
        self.lr_kernel_dev.fill_(self.lr_kernel)

    def _pass(self, x, t, arch, update=True):
        for p in self.kparams:
            p.requires_grad_(not arch)
        for p in self.aparams:
            p.requires_grad_(arch)
        if arch:
            self.agrad.zero_()  # alpha gradients arrive through autograd accumulation (softmax backward)
        with K.step_context(self.ctx):
            self.ctx.pack_all()
            loss = _loss_of(self.model, self.loss_fn, x, t)
            prev, _fused.REUSE_GRAD_OUTPUT = _fused.REUSE_GRAD_OUTPUT, True
            try:
                loss.backward(self._one)  # seed gradient kept resident: no fill launch per step
            finally:
                _fused.REUSE_GRAD_OUTPUT = prev
            self.ctx.flush_final()
        if not self.ctx.frozen and not arch:
            self.ctx.freeze()
        if not update:
            return loss.detach()
        if arch:
            K.adam_step(self.aflat, self.agrad, self.a_m, self.a_v, self.a_step, self.lr_shell, self.betas[0], self.betas[1], self.eps,
                        lr_dev=self.lr_shell_dev)
        else:
            self.fp.adam(self.lr_kernel, self.betas, self.eps, lr_dev=self.lr_kernel_dev)
        return loss.detach()

    def _both(self, x, t, vx, vt, update=True):
        la = self._pass(vx, vt, True, update)
        lw = self._pass(x, t, False, update)
        return la, lw

    def step(self, x, t, val_x, val_t):
        """returns (architecture-pass loss, weight-pass loss) as device scalars"""
        if not self.use_graph:
            return self._both(x, t, val_x, val_t)
        if self._graph is None:
            self._sx, self._st, self._svx, self._svt = x.clone(), t.clone(), val_x.clone(), val_t.clone()
            # warm-up on a side stream (allocator + lazy module state) WITHOUT the optimizer launches: weights, alphas,
            # Adam moments and step counters -- possibly just loaded from a checkpoint (search.py:108-127) -- stay untouched
            s = torch.cuda.Stream(device=self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._both(self._sx, self._st, self._svx, self._svt, update=False)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, capture_error_mode="thread_local"):  # RCCL's watchdog thread may touch the runtime meanwhile
                self._losses = self._both(self._sx, self._st, self._svx, self._svt)
            self._graph = g
        if any(a.shape != b.shape for a, b in ((x, self._sx), (t, self._st), (val_x, self._svx), (val_t, self._svt))):
            return self._both(x, t, val_x, val_t)   # remainder batch of an epoch: eager step (the graph is for one shape)
        self._sx.copy_(x); self._st.copy_(t); self._svx.copy_(val_x); self._svt.copy_(val_t)
        self._graph.replay()
        return self._losses
