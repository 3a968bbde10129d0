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

import torch
import torch.distributed as dist

from . import kernels as K
from .loss import WeightedDiceLoss


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

    def adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0):
        K.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step, lr, betas[0], betas[1], eps,
                    weight_decay, grad_scale, True)


class Trainer:
    """One searched-net (or any model built from nas_3d_unet_amd ops) training step.

    step(x, t): x (B,4,S,S,S), t (B,3,S,S,S) fp32 device tensors -> loss (0-d device tensor,
    no host sync).  With graph=True the first call captures, later calls replay."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph=True, process_group=None,
                 n_buckets=2, params=None):
        self.model = model
        self.loss_fn = WeightedDiceLoss()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(model.parameters()).device
        plist = list(params) if params is not None else list(model.parameters())
        self.fp = FlatParams(plist, self.device)
        self.use_graph = graph
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        self.n_buckets = max(1, n_buckets)
        self._graph = None
        self._static_x = self._static_t = self._static_loss = None
        self._comm_stream = torch.cuda.Stream(device=self.device) if self.world > 1 else None
        self.ctx = K.StepContext(self.device)  # batched weight packing + deferred wgrad reductions
        if self.world > 1:
            dist.broadcast(self.fp.flat, src=0, group=self.pg)

    # -- pieces ---------------------------------------------------------------------------------
    def _fwd_bwd(self, x, t):
        with K.step_context(self.ctx):
            self.ctx.pack_all()            # one launch packs every conv weight for this step
            p = self.model(x)
            loss = self.loss_fn(p, t)
            loss.backward()
            self.ctx.flush_final()         # one launch finishes every weight-gradient reduction
        if not self.ctx.frozen:
            self.ctx.freeze()              # first pass only recorded which weights / layouts are needed
        return loss.detach()

    def _allreduce(self):
        """Bucketed all-reduce of the flat gradient buffer on the side stream (xGMI ring/tree is
        latency-bound at these sizes (2-7 MB): few large buckets, not one per tensor)."""
        n = self.fp.numel
        nb = self.n_buckets
        edges = [n * i // nb // 4 * 4 for i in range(nb)] + [n]
        cs = self._comm_stream
        cs.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cs):
            for i in range(nb):
                dist.all_reduce(self.fp.grad[edges[i]:edges[i + 1]], op=dist.ReduceOp.SUM, group=self.pg)
        torch.cuda.current_stream().wait_stream(cs)

    def _update(self):
        self.fp.adam(self.lr, self.betas, self.eps, 0.0, 1.0 / self.world)

    # -- public ---------------------------------------------------------------------------------
    def step(self, x, t):
        if not self.use_graph:
            loss = self._fwd_bwd(x, t)
            if self.world > 1:
                self._allreduce()
            self._update()
            return loss
        if self._graph is None:
            self._capture(x, t)
        self._static_x.copy_(x)
        self._static_t.copy_(t)
        self._graph.replay()
        if self.world > 1:
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
        with torch.cuda.graph(g):
            self._static_loss = self._fwd_bwd(self._static_x, self._static_t)
            if self.world == 1:
                self._update()
        self._graph = g
