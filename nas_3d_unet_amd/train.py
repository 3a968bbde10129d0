"""Train-step / search-step harness: the build's counterpart of the reference's hot loops
(train.py:117-128: zero_grad -> forward -> Dice -> backward -> Adam.step;
 search.py:211-238: architecture step on the validation batch, then weight step on the train batch).

MI355X-first mechanics (none of which the reference has):
  * all kernel weights live in ONE flat fp32 buffer, their gradients in a second one; backward
    kernels write gradients in place (no per-tensor accumulate launches) and ONE fused Adam
    launch updates everything (n3d_adam_step);
  * forward + backward (+ Adam) of a fixed-shape step is captured once into a HIP graph and
    replayed, which removes the per-kernel host launch cost (~300 launches per step);
  * data parallel: one process per GPU, parameters broadcast once, per step a SUM all-reduce of the flat gradient buffer over
    RCCL through the C ABI's n3d_comm_* on ONE communicator per process (comm.py: no torch.distributed collective inside the step,
    so nothing of ProcessGroupNCCL's watchdog can meet a stream capture), the mean folded into the Adam kernel.  Default: ONE
    bucket issued from the step's own stream behind the step's graphs.  n_buckets >= 2: on the side-stream schedule a bucket that
    the backward walk has completed is reduced and all-reduced ON THE WEIGHT-GRADIENT STREAM under the rest of the backward
    (device flags, no host wait); without a side schedule the exchange stays a single bucket.
"""
from __future__ import annotations

import os
import types
import weakref

import torch
import torch.distributed as dist

from . import comm as _comm
from . import fused as _fused
from . import kernels as K
from .loss import WeightedDiceLoss


def _loss_of(model, loss_fn, x, t):
    """Dice loss of the model on (x, t): through the model's fused head + loss path when it has one"""
    fl = getattr(model, "forward_loss", None)
    if fl is not None and isinstance(loss_fn, WeightedDiceLoss):
        return fl(x, t, loss_fn.smooth)[0]
    return loss_fn(model(x), t)


GRAD_HEADER = 4   # floats in front of a flat gradient buffer (FlatParams.grad_full)


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
        # 16 bytes in front of the gradients: word 0 is the "a stream hand-off of this rank timed out" flag that data-parallel
        # ranks SUM-all-reduce together with the gradients (GradSync), so that every rank withholds the same update
        self.grad_full = torch.zeros(n + GRAD_HEADER, dtype=torch.float32, device=device)
        self.grad = self.grad_full[GRAD_HEADER:]
        self.exp_avg = torch.zeros(n, dtype=torch.float32, device=device)
        self.exp_avg_sq = torch.zeros(n, dtype=torch.float32, device=device)
        self.step = K.step_counter(device) if torch.device(device).type == "cuda" else torch.zeros(1, dtype=torch.int32, device=device)
        self.offsets = offs
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                v = self.flat[o:o + p.numel()].view(p.shape)
                v.copy_(p.data)
                p.data = v
                g = self.grad[o:o + p.numel()].view(p.shape)
                p._n3d_grad = g  # backward kernels write here (programs.py / kernels.grad_target)
                p.grad = g

    def adam(self, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, grad_scale=1.0, lr_dev=None, guard=None):
        K.adam_step(self.flat, self.grad, self.exp_avg, self.exp_avg_sq, self.step, lr, betas[0], betas[1], eps,
                    weight_decay, grad_scale, True, lr_dev, guard)


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


def _plateau_for(owner, lr_attr, setter):
    """PlateauLR bound to a trainer through a weak reference: no trainer <-> scheduler cycle, so a dropped trainer (and the
    HIP graphs / private memory pool it owns) is released at once by reference counting, never by a cyclic collection that could
    fire in the middle of another trainer's graph capture"""
    ref = weakref.ref(owner)
    return PlateauLR(lambda: getattr(ref(), lr_attr), lambda v: getattr(ref(), setter)(v))


class GradSync:
    """Bucketed SUM all-reduce of one flat gradient buffer; callers divide by the world size (folded into the Adam kernel).

    xGMI is point-to-point (7 links per GPU) and the whole payload is 2-7 MB (searched net) / 27 MB (supernet), so the
    collective is latency-bound: a few large buckets, never one call per tensor.  `ranges`: the buckets as (begin, end)
    element ranges in ISSUE order (default: n_buckets equal slices).
    backend (default: "rccl" for CUDA tensors on an NCCL group, else "torch"):
      "rccl"  = n3d_comm_allreduce_sum on the process's shared communicator (comm.for_group): stream-ordered on the CURRENT stream,
                no Work object, no event, no thread that polls it later -- nothing that can collide with a stream capture
                (comm.py says what did in round 4);
      "torch" = torch.distributed.all_reduce (gloo on CPU tensors; two gloo processes sharing one GPU).  On an NCCL group the call is
                issued on a comm stream of this object's own that is NEVER captured, tied to the current stream by two event waits:
                ProcessGroupNCCL's watchdog polls the Work's end event until it has retired it, and HIP refuses that poll while the
                stream the event was last recorded on is capturing."""

    def __init__(self, flat_grad, process_group=None, n_buckets=2, ranges=None, backend=None, header=None):
        """header: the full buffer `flat_grad` is a view of (FlatParams.grad_full: GRAD_HEADER floats in front of it) -- those
        words are exchanged together with the bucket that is issued LAST (the one that starts at 0)"""
        self.header = header
        self.g = flat_grad
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        n = flat_grad.numel()
        nb = max(1, min(n_buckets, n // 4 if n >= 4 else 1))
        self.edges = [n * i // nb // 4 * 4 for i in range(nb)] + [n]
        self.ranges = list(ranges) if ranges is not None else list(zip(self.edges[:-1], self.edges[1:]))
        self.force = dist.is_initialized() and os.environ.get("N3D_FORCE_DP") == "1"  # see Trainer.dp_path
        self.active = self.world > 1 or self.force
        self._comm = None
        self._torch_stream = None
        want = backend or os.environ.get("N3D_COMM") or "rccl"
        if want not in ("rccl", "torch"):
            raise ValueError("GradSync: backend must be 'rccl' or 'torch', not %r" % (want,))
        if self.active and want == "rccl":
            self._comm = _comm.for_group(self.pg, flat_grad.device)     # None: CPU tensors / not an NCCL group -> torch.distributed
        self.backend = "rccl" if self._comm is not None else "torch"
        if self.active and self.backend == "torch" and flat_grad.is_cuda and _comm._is_nccl(self.pg):
            self._torch_stream = torch.cuda.Stream(device=flat_grad.device)

    def close(self):
        """(the communicator is the process's shared one: comm.close_all() destroys it)"""
        self._comm = None

    def reduce_range(self, i):
        """SUM all-reduce of bucket i, ordered on the CURRENT stream"""
        if not self.active:
            return
        a, b = self.ranges[i]
        if b <= a:
            return
        buf = self.g
        if self.header is not None and a == 0 and i == len(self.ranges) - 1:
            buf, b = self.header, b + GRAD_HEADER      # the hand-off flag rides in front of the last bucket
        if self._comm is not None:
            self._comm.allreduce_sum_ptr(buf.data_ptr() + 4 * a, b - a)
        elif self._torch_stream is not None:
            cur, cs = torch.cuda.current_stream(), self._torch_stream
            cs.wait_stream(cur)
            with torch.cuda.stream(cs):
                dist.all_reduce(buf[a:b], op=dist.ReduceOp.SUM, group=self.pg)
            cur.wait_stream(cs)
        else:
            dist.all_reduce(buf[a:b], op=dist.ReduceOp.SUM, group=self.pg)

    def all_reduce(self):
        """sum-reduce every bucket, in issue order"""
        if not self.active:
            return
        for i in range(len(self.ranges)):
            self.reduce_range(i)

    def broadcast(self, t, src=0):
        """rank `src`'s copy of t to every rank (once, when a trainer is built)"""
        if self.world <= 1:
            return
        if self._comm is not None:
            self._comm.broadcast(t, src)
        else:
            dist.broadcast(t, src=dist.get_global_rank(self.pg, src) if self.pg is not None else src, group=self.pg)

    def all_true(self, flag):
        """True only if `flag` is true on every rank (data-parallel ranks must take the same schedule decisions: they decide which
        graphs exist and in which order collectives are issued)"""
        if self.world <= 1:
            return bool(flag)
        if self._comm is not None:
            return self._comm.all_true(flag)
        ok = torch.tensor([1.0 if flag else 0.0], device=self.g.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN, group=self.pg)
        return float(ok.item()) != 0.0


_capture_streams = {}


def _dev_key(device):
    """index of a torch device; an index-less 'cuda' means the current device"""
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def capture_stream(device):
    """the ONE stream every trainer of this process warms up and captures on.  Streams are hardware queues: with more of them than
    the scheduler keeps resident side by side (four, counting the default stream, on this stack) a device-side wait of the side
    schedule is only relieved at the end of a time slice -- so nothing here makes a stream it does not need"""
    key = _dev_key(device)
    st = _capture_streams.get(key)
    if st is None:
        st = _capture_streams[key] = torch.cuda.Stream(device=device)
    return st


_side_streams = {}   # device index -> [side streams made so far]: trainers share them (every new HIP stream may become another
                     # hardware queue, and with more queues than the scheduler keeps resident a spinning wait kernel is only
                     # relieved at the end of a ~1 ms time slice)


def _low_priority_stream(device):
    """a stream of the LOWEST priority the runtime offers (the side work must not take compute units from the backward chain);
    torch only exposes normal / high, so the stream is made through HIP and wrapped"""
    try:
        with torch.cuda.device(device):
            # (A CU mask on the side stream does not help: hipExtStreamCreateWithCUMask makes a BLOCKING stream -- next to torch's
            # legacy default stream every hand-off then takes a time slice, a 9.6 ms step -- and with the whole step moved to a
            # non-blocking stream, a side stream held to 28 or 24 compute units of every XCD [mask bit i = CU i // 8 of XCD i % 8,
            # tools/cumask_map.cpp; a mask that empties an XCD is ignored] leaves the chain's time under the side work where it
            # was, 2.01 ms: what the side work costs the chain is not compute units.  profiles/r03_contention_probes.log)
            h = K.stream_create_low_priority()     # through libn3d: the HIP runtime that launches the kernels makes the stream
        return torch.cuda.ExternalStream(h, device=device)
    except K.N3DError:
        return torch.cuda.Stream(device=device)


def reserve_side_streams(device, n=2):
    """Make the process's side streams NOW and put a launch on each.  Which hardware queue / pipe a HIP stream gets is settled when
    it is first used, in creation order; streams made early sit next to the default stream's queue and run side by side with it,
    while streams made after a lot of other activity were seen to share a time slice with it (a search step of 58 instead of
    14.5 ms; docs/history/DESIGN_r01-r03.md).  Every trainer calls this first; a program that does other GPU work before it builds a
    trainer can call it right after selecting the device.  (A late reservation is not an error: the trainers time both schedules
    and keep the single-stream one if the side streams do not pay.)"""
    device = torch.device(device)
    if device.type != "cuda":
        return []
    pool = _side_streams.setdefault(_dev_key(device), [])
    made = False
    while len(pool) < n:
        pool.append(_low_priority_stream(device))
        made = True
    if made:
        scratch = torch.zeros(8, dtype=torch.int32, device=device)
        scratch[0] = 1
        for st in pool:
            with K.on_side(st):
                K.sync_signal(scratch.data_ptr() + 16, scratch.data_ptr(), False)
        torch.cuda.synchronize(device)
    return pool


class SideSchedule:
    """Weight-gradient kernels on a SIDE HIP stream, tied to the backward chain by flags in device memory (include/n3d.h,
    "stream hand-off"; round 3).

    Nothing on the backward chain waits for a weight gradient until the slab reduction in front of Adam, and the chain is
    latency-bound (a few workgroups per launch), so the chip has room for them next to it.  With `deferring()` active the
    weight-gradient launches are queued (kernels.StepContext.wq); at a cut point -- the end of a node / cell of the backward
    walk with at least `min_queue` launches queued, and the end of the walk -- the main stream stores the step number to a
    flag and the queue gets a mark.  `launch_side()` (on the side stream) then issues, per mark, a one-lane wait kernel on
    that flag followed by the queued kernels, and publishes a 'done' flag; `finish()` (main stream) waits for it and runs the
    slab reduction.  Main chain, side work and tail are replayed as three separately launched HIP graphs: an event between
    graph launches costs ~190 us per hand-off on this stack and an intra-graph fork ~19 us per edge without overlapping, a
    flag costs a ~2 us kernel on each side (tools/handoff_cost.cpp).
    Device words of `sync` (int32): [0] main-stream step, [1] time-outs, [2] side-stream step, [3] weight-gradient stream step,
    [4] time-outs the host has acknowledged, [8 + i] flag of cut i, [8 + JOIN] side work of this step done.

    A wait that gives up lets its stream go on (never a hung GPU), so the step's gradients may be wrong.  That must not reach the
    weights: every optimizer launch of a trainer with a SideSchedule is GUARDED (`guard()`, n3d_adam_step_guarded) -- while
    sync[1] != sync[4] it updates nothing, writes NaN to the loss and sets a host-visible word.  The trainers poll that word at
    every step() (no HIP call) and at every host-visible point (check()); a time-out is FATAL for the schedule: the trainer
    raises until `acknowledge()` (Trainer.recover) has been called, which also retires the side schedule."""
    JOIN = 400

    def __init__(self, device, ctx, min_queue=None, wgrad_stream=False):
        """wgrad_stream: a SECOND side stream for the weight-gradient launches (the supernet's weight pass keeps the first one busy
        with the preprocess-fed edges of its forward and backward); without it, or when no second stream passes the probe, the
        weight gradients share the one side stream"""
        self.device, self.ctx = device, ctx
        self.min_queue = 5 if min_queue is None else int(min_queue)
        # cuts from the end at which the side stream reduces the slabs it has so far (0 = off, the default: measured at 64^3 the
        # tail shrinks 66 -> 49 us but the reduction takes 24 us out of the chain it runs beside)
        self.early_finalize = 0
        # live cuts: the slabs launched so far are reduced on the weight-gradient stream behind the n-th group (0 = never, -1 = at 55 %
        # of the groups the fullest pass so far had).  64^3 train step: tail 0.078 -> 0.064 ms, chain +0.001 (n = 11 of 20)
        self.early_at = -1
        # weight-gradient groups, counted from the end of the walk, that go to the INLINE side stream instead (see cut()).  (Every k-th
        # group there, or a second weight-gradient stream, measured slower: profiles/r04_alternate_ab.log)
        self.tail_inline = (2,)
        self._live_cuts = 0
        self._live_cuts_max = 0
        self.sync = torch.zeros(8 + self.JOIN + 8, dtype=torch.int32, device=device)
        self.sync[0] = 1
        self.sync[2] = 1
        self.host_word = K.HostWord()     # set by a guarded update that was withheld
        self.failed = 0                   # time-outs seen and not yet acknowledged (the trainers refuse to step while > 0)
        self._cuts = 0
        self._main_jobs = {}
        self._last_cut = -1
        self._pass_active = False
        self.live = False      # backward_mode(): cuts hand their group to the side stream at once
        self.bwd_side_inputs = (0, 1)   # backward_mode(): the side stream takes the edges fed by these preprocess outputs
        self.arena = []        # every tensor made while forward_mode() / arena_mode() is on: held until finish()
        self.seen = 0
        self.replays = 0
        self.watch = None
        # N3D_SIDE_TRACE=1 (tools/side_timeline.py): wall-clock stamps next to every hand-off.  int64 words: [0] main: first cut,
        # [2 i + 2] main stored flag i, [2 i + 3] side passed wait i, [2 JOIN + 4] side done, [2 JOIN + 5] main passed the join,
        # [2 JOIN + 6] slab reduction launched behind it, [2 JOIN + 8 + i] the weight-gradient group behind cut i has been launched through
        # (its last kernel is done when the stamp executes: group busy time = this minus [2 i + 3])
        self.trace = torch.zeros(3 * self.JOIN + 16, dtype=torch.int64, device=device) if os.environ.get("N3D_SIDE_TRACE") == "1" else None
        self.stream = self._probe()
        self.wstream = self.stream
        if wgrad_stream and self.stream is not None:
            w = self._probe(exclude=(self.stream,))
            if w is not None:
                self.wstream = w
        self.sync[3] = 1          # the weight-gradient stream's own step counter (used when it is a stream of its own)
        self._side_tok = None     # the side stream's latest flag of this pass (live cuts make the weight-gradient stream wait for it)

    @property
    def split(self):
        return self.wstream is not self.stream

    def wptr(self):
        """step word of the stream that runs the weight gradients"""
        return self.ptr(3) if self.split else self.ptr(2)

    def ptr(self, i):
        return self.sync.data_ptr() + 4 * i

    def _probe(self, exclude=()):
        """a side stream that really runs next to the current one: a few device-side ping-pongs must complete without a time-out
        and fast (a wait kernel at the head of the hardware queue that also carries its signal only ends by its time-out; streams
        that share a time-sliced queue hand over in ~1 ms instead of ~5 us).  Tries a few streams; None = schedule off."""
        import time
        main = torch.cuda.current_stream(self.device)
        rounds = 8
        pool = _side_streams.setdefault(_dev_key(self.device), [])
        for attempt in range(5):
            if attempt < len(pool):
                side = pool[attempt]         # a stream an earlier trainer of this process made: reuse before making another
            else:
                side = _low_priority_stream(self.device) if attempt < 3 else torch.cuda.Stream(device=self.device)
                pool.append(side)
            if any(side is e for e in exclude):
                continue
            ok = True
            for timed in (False, True):      # the first pass also loads the two kernels
                probe = torch.zeros(4 + 2 * rounds, dtype=torch.int32, device=self.device)   # [0] step = 1, [1] time-outs, [4 + i] flags
                probe[0] = 1
                base = probe.data_ptr()
                torch.cuda.synchronize(self.device)
                t0 = time.perf_counter()
                with torch.cuda.stream(side):
                    for i in range(rounds):     # side waits for main's flag i, then publishes its own
                        K.sync_wait(base + 4 * (4 + 2 * i), base, base + 4, False, 20000)   # <= ~10 ms each
                        K.sync_signal(base + 4 * (5 + 2 * i), base, False)
                with torch.cuda.stream(main):
                    for i in range(rounds):
                        K.sync_signal(base + 4 * (4 + 2 * i), base, False)
                        K.sync_wait(base + 4 * (5 + 2 * i), base, base + 4, False, 20000)
                torch.cuda.synchronize(self.device)
                dt = time.perf_counter() - t0
                ok = ok and int(probe[1].item()) == 0 and (not timed or dt < 5e-3)
            if ok:
                return side
        return None

    def guard(self, loss=None, peer_flag=None):
        """UpdateGuard for an optimizer launch behind this schedule's hand-offs (loss: the scalar that turns NaN)"""
        return K.UpdateGuard(self.ptr(1), self.ptr(4), peer_flag, loss.data_ptr() if loss is not None else None, self.host_word.ptr)

    def _fail(self, n):
        raise K.N3DError("side-stream schedule: %d device-side wait(s) timed out -- the streams did not run concurrently.  The updates of "
                         "the affected steps were WITHHELD on the device (weights, Adam moments and step counters are those of the last "
                         "good step; the loss of a withheld step reads NaN).  The trainer refuses to step until recover() is called, "
                         "which acknowledges the time-outs and continues on one stream (N3D_SIDE_WGRAD=0 turns the schedule off)" % n)

    def poll(self):
        """host-only test (no HIP call, no synchronisation): raises if a guarded update has been withheld, or if an earlier
        check() found time-outs that nobody has acknowledged"""
        if self.failed or self.host_word.value:
            if not self.failed:
                self.failed = max(1, int(self.sync[1].item()) - self.seen)
            self._fail(self.failed)

    def check(self):
        """synchronising test: raises if a device-side wait has timed out since the last acknowledge()"""
        n = int(self.sync[1].item())
        if n != self.seen:
            self.failed = n - self.seen
        if self.failed or self.host_word.value:
            self._fail(max(self.failed, 1))

    def sync_timeouts_now(self):
        """the device's time-out counter (synchronises)"""
        return int(self.sync[1].item())

    def acknowledge(self):
        """the caller has dealt with the time-outs (Trainer.recover): updates are allowed again"""
        torch.cuda.synchronize(self.device)
        self.sync[4] = self.sync[1]
        self.seen = int(self.sync[1].item())
        self.failed = 0
        self.host_word.clear()
        torch.cuda.synchronize(self.device)

    class _Deferring:
        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            o = self.o
            if not o._pass_active:
                o.begin_pass()
            o.ctx.defer_wgrad = True
            self.prev = (_fused.CELL_DONE_HOOK, _fused.NODE_DONE_HOOK)
            outer = self.prev[0]

            def cell(k):
                o.cut(final=k < 0)
                if outer is not None:
                    outer(k)
            _fused.CELL_DONE_HOOK, _fused.NODE_DONE_HOOK = cell, o.cut
            return o

        def __exit__(self, *exc):
            if exc[0] is None:
                self.o.cut(final=True)     # whatever was queued after the last cut of the walk (still on the main stream)
            self.o.ctx.defer_wgrad = False
            _fused.CELL_DONE_HOOK, _fused.NODE_DONE_HOOK = self.prev
            return False

    # -- one pass (forward + backward of one batch) = one step of the flag protocol ------------------------------------------
    def begin_pass(self):
        """flag ids are handed out from 0 within a pass (forward hand-offs and the cuts of the backward walk share them)"""
        self._cuts = 0
        self._live_cuts = 0
        self._main_jobs = {}
        self._last_cut = -1
        self._side_tok = None
        self._pass_active = True

    def _flag(self):
        i = self._cuts
        if i >= self.JOIN:
            raise K.N3DError("side-stream schedule: more than %d hand-offs in one pass" % self.JOIN)
        self._cuts += 1
        return i

    # -- inline side work of the forward pass (fused._run_forward_side) -------------------------------------------------------
    def fork(self):
        """main stream: store a flag behind everything launched so far; returns its id"""
        i = self._flag()
        K.sync_signal(self.ptr(8 + i), self.ptr(0), False)
        return i

    class _Side:
        def __init__(self, owner, wait_id):
            self.o, self.wait_id = owner, wait_id

        def __enter__(self):
            self.redirect = K.on_side(self.o.stream)
            self.redirect.__enter__()
            K.sync_wait(self.o.ptr(8 + self.wait_id), self.o.ptr(2), self.o.ptr(1), False)
            return self

        def __exit__(self, *exc):
            return self.redirect.__exit__(*exc)

    def side(self, wait_id):
        """with side(flag id): launches go to the side stream, behind a device-side wait on that flag of the main stream"""
        return SideSchedule._Side(self, wait_id)

    def side_signal(self):
        """inside side(): store a flag behind the side stream's launches so far; returns its id"""
        i = self._flag()
        K.sync_signal(self.ptr(8 + i), self.ptr(2), False)
        if self.trace is not None:
            K.stamp(self.trace.data_ptr() + 8 * (2 * i + 2))      # (a join flag: the side stream stored it ...)
        self._side_tok = i
        return i

    def join(self, i):
        """main stream: wait for a flag of the side stream"""
        if self.trace is not None:
            K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 8 + i))   # (... the main stream arrived at its wait ...)
        K.sync_wait(self.ptr(8 + i), self.ptr(0), self.ptr(1), False)
        if self.trace is not None:
            K.stamp(self.trace.data_ptr() + 8 * (2 * i + 3))      # (... and got past it)

    def join_many(self, ids):
        """main stream: wait for several flags of the side stream -- two per launch (n3d_sync_wait2)"""
        ids = list(ids)
        k = 0
        while k + 1 < len(ids):
            a, b = ids[k], ids[k + 1]
            if self.trace is not None:
                for i in (a, b):
                    K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 8 + i))
            K.sync_wait2(self.ptr(8 + a), self.ptr(8 + b), self.ptr(0), self.ptr(1), False)
            if self.trace is not None:
                for i in (a, b):
                    K.stamp(self.trace.data_ptr() + 8 * (2 * i + 3))
            k += 2
        if k < len(ids):
            self.join(ids[k])

    class _ArenaMode:
        """every tensor created inside (on the calling thread) is held in owner.arena until finish()"""

        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            from torch.utils._python_dispatch import TorchDispatchMode
            keep = self.o.arena

            class Arena(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    out = func(*args, **(kwargs or {}))
                    if isinstance(out, torch.Tensor):
                        keep.append(out)
                    elif isinstance(out, (tuple, list)):
                        keep.extend(t for t in out if isinstance(t, torch.Tensor))
                    return out

            self.mode = Arena()
            self.mode.__enter__()
            return self

        def __exit__(self, *exc):
            return self.mode.__exit__(*exc)

    def arena_mode(self):
        return SideSchedule._ArenaMode(self)

    class _Backward:
        """backward_mode(): supernet cells run the backward of their preprocess-fed edges on the side stream (fused.SIDE_BWD), and
        the schedule is LIVE: a cut of the backward walk hands the weight-gradient launches queued so far to the side stream at
        once (behind a wait on the cut's flag) instead of at the end of the pass -- the side stream is busy during the walk now,
        what is appended at the end would run after all of it"""

        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            if not self.o._pass_active:
                self.o.begin_pass()
            self.prev, _fused.SIDE_BWD = _fused.SIDE_BWD, self.o
            self.o.live = True
            return self.o

        def __exit__(self, *exc):
            _fused.SIDE_BWD = self.prev
            self.o.live = False
            return False

    def backward_mode(self, side_inputs=(0, 1)):
        self.bwd_side_inputs = tuple(side_inputs)
        return SideSchedule._Backward(self)

    class _Forward:
        """forward_mode(): supernet cells run their off-chain edges on the side stream (fused.SIDE_FWD), and every tensor created
        meanwhile is held in `arena` until finish().  Allocations stay on torch's current stream; with two streams in flight the
        allocator's stream-ordered reuse of a block freed by Python is not ordered against the OTHER stream's pending kernels, so
        nothing made during the pass is released before the streams have joined (a TorchDispatchMode sees every new tensor)."""

        def __init__(self, owner):
            self.o = owner

        def __enter__(self):
            from torch.utils._python_dispatch import TorchDispatchMode
            keep = self.o.arena

            class Arena(TorchDispatchMode):
                def __torch_dispatch__(self, func, types, args=(), kwargs=None):
                    out = func(*args, **(kwargs or {}))
                    if isinstance(out, torch.Tensor):
                        keep.append(out)
                    elif isinstance(out, (tuple, list)):
                        keep.extend(t for t in out if isinstance(t, torch.Tensor))
                    return out

            if not self.o._pass_active:
                self.o.begin_pass()
            self.mode = Arena()
            self.mode.__enter__()
            self.prev, _fused.SIDE_FWD = _fused.SIDE_FWD, self.o
            return self.o

        def __exit__(self, *exc):
            _fused.SIDE_FWD = self.prev
            return self.mode.__exit__(*exc)

    def forward_mode(self):
        return SideSchedule._Forward(self)

    # -- the side stream's work as a HIP graph captured next to torch's capture of the main stream --------------------------------
    def close_bucket(self, on_closed):
        """data parallel, bucketed exchange on device flags: every gradient of a bucket has been LAUNCHED (the backward walk is past
        the cell that completes it).  The weight-gradient stream takes whatever is still queued, waits for the main stream's flag
        behind everything launched so far (and the side stream's latest flag), reduces every slab launched so far -- and
        `on_closed()` then puts the bucket's all-reduce on that stream (eagerly, or as the boundary between two captured segments
        of the stream).  The chain does not wait for anything here."""
        self.cut(final=True)
        i = self.fork()
        with K.on_side(self.wstream):
            word = self.wptr()
            K.sync_wait(self.ptr(8 + i), word, self.ptr(1), False)
            if self.split and self._side_tok is not None:
                K.sync_wait(self.ptr(8 + self._side_tok), word, self.ptr(1), False)
            with K.step_context(self.ctx):
                self.ctx.finalize_now(len(self.ctx.final))
        on_closed()

    def split_wstream_capture(self):
        """end the capture segment of the weight-gradient stream and open the next one (the all-reduce of a closed bucket is
        launched between the two at replay)"""
        st = self.wstream
        self._wsegs.append(K.stream_capture_end(st.cuda_stream))
        K.stream_capture_begin(st.cuda_stream)

    def raw_capture_begin(self):
        """the side stream starts capturing (thread-local mode, like torch's capture of the main stream); launches redirected to it
        while the main stream is being captured land in a graph of their own.  Capture, instantiation and launch go through
        libn3d (n3d_stream_capture_* / n3d_graph_*): the HIP runtime that launches the kernels owns the graphs."""
        self._wsegs = []
        for st in ([self.stream, self.wstream] if self.split else [self.stream]):
            K.stream_capture_begin(st.cuda_stream)

    def raw_capture_end(self):
        """-> [(executable graph handle, stream)] (replayed with raw_replay, released with raw_destroy).  The weight-gradient stream
        may have been captured in several segments (split_wstream_capture): its entries then come in segment order, and
        `self.wseg_count` says how many of the returned entries (the last ones) they are."""
        out, err = [], None
        segs, self._wsegs = self._wsegs, []
        for st in ([self.stream, self.wstream] if self.split else [self.stream]):
            try:
                ex = K.stream_capture_end(st.cuda_stream)
                if st is self.wstream:
                    out.extend((e, st) for e in segs + [ex])
                else:
                    out.append((ex, st))
            except K.N3DError as e:      # end the other stream's capture too before raising
                err = e
        self.wseg_count = len(segs) + 1
        if err is not None:
            self.raw_destroy(out)
            raise err
        return out

    def raw_replay(self, execs):
        for ex, st in execs:
            K.graph_launch(ex, st.cuda_stream)

    @staticmethod
    def raw_destroy(execs):
        for ex, _ in execs or ():
            K.graph_destroy(ex)

    def deferring(self):
        """with side.deferring(): run forward + backward; weight-gradient launches are queued, flags stored at the cut points"""
        return SideSchedule._Deferring(self)

    def cut(self, final=False):
        """a cut point of the backward walk; final (or force) = cut whatever is queued, however little"""
        ctx = self.ctx
        n = ctx.queued()
        if n > 0 and (final or (n >= self.min_queue and self._cuts < self.JOIN - 1)):
            i = self._flag()
            K.sync_signal(self.ptr(8 + i), self.ptr(0), False)
            self._main_jobs[i] = len(ctx.final)     # slab-reduction jobs the main chain has issued in front of this flag
            if self.trace is not None:
                K.stamp(self.trace.data_ptr() + 8 * (2 * i + 2))
            if self.live:
                # the side stream takes the group now: wait for the flag, then the launches (their slab-reduction jobs are recorded)
                # the weight-gradient stream is backlogged at the end of the walk (the 64^3-level kernels: the join waited ~45 us
                # behind the chain's last cut, tools/side_timeline.py) while the inline side stream is mostly idle there: the group
                # tail_inline positions from the end (default: the second to last) goes to that stream instead; 1.865 -> 1.85 ms
                inline = self.split and self._live_cuts_max >= 8 and (self._live_cuts_max - self._live_cuts) in self.tail_inline
                with K.on_side(self.stream if inline else self.wstream):
                    word = self.ptr(2) if inline else self.wptr()
                    K.sync_wait(self.ptr(8 + i), word, self.ptr(1), False)
                    if self.split and self._side_tok is not None and not inline:
                        # operands the SIDE stream produced (the d(raw) of its edges): behind its latest flag
                        K.sync_wait(self.ptr(8 + self._side_tok), word, self.ptr(1), False)
                    if self.trace is not None:
                        K.stamp(self.trace.data_ptr() + 8 * (2 * i + 3))
                    ctx.flush_wgrads(inline=inline)
                    if self.trace is not None:
                        K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 8 + i))
                    self._live_cuts += 1
                    self._live_cuts_max = max(self._live_cuts_max, self._live_cuts)
                    at = self.early_at if self.early_at >= 0 else (int(round(0.55 * self._live_cuts_max)) if self._live_cuts_max >= 8 else 0)
                    if at and self._live_cuts == at and not inline:   # (an inline group runs on the OTHER side stream: the slabs of
                        # the weight-gradient stream's launches are not ordered in front of a reduction issued there)
                        # the slabs of everything launched so far (and of the chain's own jobs in front of this flag) are reduced on
                        # the weight-gradient stream here: the reduction behind the join only has the last groups left
                        ctx.finalize_now(self._main_jobs[i])
                return
            ctx.wq.insert(len(ctx.wq) - n, ("mark", i))    # the wait goes IN FRONT of the launches it guards
            ctx.wq.append(("mark", -1))                    # closes the group (no wait)
            self._last_cut = i

    def launch_side(self, redirect=False):
        """the weight-gradient stream: per group a device-side wait for the main stream's flag, then the queued launches; then the
        'done' flags of this pass (one per side stream).  redirect=False: the caller has made the (single) side stream current;
        True: the launches are redirected here (K.on_side), as the search trainer's simultaneous captures need"""
        marks = [it[1] for it in self.ctx.wq if it[0] == "mark" and it[1] >= 0]
        early = marks[-self.early_finalize] if 0 < self.early_finalize < len(marks) else -1
        prev = {t: (marks[k - 1] if k > 0 else None) for k, t in enumerate(marks)}

        def on_mark(tag):
            if tag >= 0:
                if tag == early and prev[tag] is not None:
                    # the side stream idles between its groups: the slabs of everything it has launched so far (and of the main
                    # chain's own jobs in front of the flag it passed last) are reduced here, so that the reduction behind the join
                    # only has the last few groups left
                    self.ctx.finalize_now(self._main_jobs[prev[tag]])
                K.sync_wait(self.ptr(8 + tag), self.wptr(), self.ptr(1), False)
                if self.trace is not None:
                    K.stamp(self.trace.data_ptr() + 8 * (2 * tag + 3))

        import contextlib
        with (K.on_side(self.wstream) if redirect else contextlib.nullcontext()):
            with K.step_context(self.ctx):
                self.ctx.flush_wgrads(on_mark)
            if self.split:
                K.sync_signal(self.ptr(8 + self.JOIN + 1), self.ptr(3), True)
        with (K.on_side(self.stream) if redirect else contextlib.nullcontext()):
            K.sync_signal(self.ptr(8 + self.JOIN), self.ptr(2), True)
            if self.trace is not None:
                K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 4))

    def finish(self):
        """on the main stream: wait for the side stream's 'done' flag, then reduce the weight-gradient slabs (one launch)"""
        if self.split:
            K.sync_wait(self.ptr(8 + self.JOIN + 1), self.ptr(0), self.ptr(1), False)
        K.sync_wait(self.ptr(8 + self.JOIN), self.ptr(0), self.ptr(1), True)
        self._pass_active = False
        self.arena.clear()        # (rebinding would orphan the list the dispatch mode of forward_mode() appends to)
        if self.trace is not None:
            K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 5))
        with K.step_context(self.ctx):
            self.ctx.flush_final()
        if self.trace is not None:
            K.stamp(self.trace.data_ptr() + 8 * (2 * self.JOIN + 6))


class _Ctx:
    """stand-in for an autograd context when a Function's forward / backward are called directly (Trainer's pipeline)"""

    def __init__(self, needs):
        self.needs_input_grad = needs

    def mark_non_differentiable(self, *a):
        pass


def flatten_params(params, device=None):
    """(flat parameter tensor, flat gradient tensor): re-homes every parameter (and its .grad) as a view."""
    params = list(params)
    device = device if device is not None else params[0].device
    offs, n = [], 0
    for p in params:
        offs.append(n)
        n += (p.numel() + 3) // 4 * 4
    flat = torch.zeros(n, dtype=torch.float32, device=device)
    full = torch.zeros(n + GRAD_HEADER, dtype=torch.float32, device=device)   # header: see FlatParams.grad_full
    grad = full[GRAD_HEADER:]
    grad._n3d_full = full
    with torch.no_grad():
        for p, o in zip(params, offs):
            v = flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = grad[o:o + p.numel()].view(p.shape)
    return flat, grad, offs


def _dropout_states(model):
    return [m._n3d_state for m in model.modules() if getattr(m, "_n3d_state", None) is not None]


def _dropout_snapshot(model, device):
    """snapshot of every Dropout3d generator of the model (created now if a module has not drawn yet)"""
    from . import programs as _P
    for m in model.modules():
        if isinstance(m, torch.nn.Dropout3d):
            _P.dropout_state(m, device)
    return _Snapshot(_dropout_states(model))


class _Snapshot:
    """optimizer-visible state of a trainer (weights, Adam moments, step counters, dropout generators): taken before the
    schedules are timed against each other on the real step, put back afterwards"""

    def __init__(self, tensors):
        self.tensors = [t for t in tensors if t is not None]
        self.saved = [t.clone() for t in self.tensors]

    def restore(self):
        with torch.no_grad():
            for t, v in zip(self.tensors, self.saved):
                t.copy_(v)


def _time_schedule(run, device, warm=2, reps=6):
    import time
    for _ in range(warm):
        run()
    torch.cuda.synchronize(device)
    t0 = time.perf_counter()
    for _ in range(reps):
        run()
    torch.cuda.synchronize(device)
    return (time.perf_counter() - t0) / reps


class _TwinShell:
    """what SearchTrainer runs for a ShellNet whose kernel net has odd channel counts: the shell's own alphas around the kernel net's
    zero-padded twin (unet.PaddedTwin)"""

    def __init__(self, shell, twin):
        self.shell, self.kernel = shell, twin

    def forward_loss(self, x, t, smooth=1e-6):
        from . import unet as _unet
        return _unet.run_loss(self.kernel, x, t, tuple(self.shell._soft()), smooth)

    def modules(self):
        return self.kernel.modules()


def _pad_mask_for(tw, real, fp, device):
    """flat 0/1 mask over a FlatParams of a padded twin's parameters: 1 at the real entries"""
    names = {id(q): n for n, q in tw.twin.named_parameters()}
    rshape = {n: q.shape for n, q in real.named_parameters()}
    mask = torch.zeros_like(fp.grad)
    for q, o in zip(fp.params, fp.offsets):
        n = names[id(q)]
        mask[o:o + q.numel()] = tw.embed_tensor(n, torch.ones(rshape[n], device=device), q.shape).reshape(-1)
    return mask


def _padded_flags():
    """context: while kernels of a padded twin are launched -- conv-bias gradients by summation, per-term GroupNorm launches (the node-level
    ones share their element count with SE gates), no node-planar inner cells (unet.run_padded does the same)"""
    import contextlib
    from . import programs as _P

    @contextlib.contextmanager
    def ctx():
        prev = (_P.ANALYTIC_CONV_BIAS, _fused.NODE_PHASES, _fused.NODE_APPLY, _P.NODE_FWD_COEFFS, _fused.PLANAR_INNER)
        _P.ANALYTIC_CONV_BIAS, _fused.NODE_PHASES, _fused.NODE_APPLY, _P.NODE_FWD_COEFFS, _fused.PLANAR_INNER = False, False, False, False, False
        try:
            yield
        finally:
            _P.ANALYTIC_CONV_BIAS, _fused.NODE_PHASES, _fused.NODE_APPLY, _P.NODE_FWD_COEFFS, _fused.PLANAR_INNER = prev
    return ctx()


class Trainer:
    """One searched-net (or any model built from nas_3d_unet_amd ops) training step.

    step(x, t): x (B,4,S,S,S), t (B,3,S,S,S) fp32 device tensors -> loss (0-d device tensor,
    no host sync).  With graph=True the first call captures, later calls replay.
    n_buckets >= 2 (data parallel only): bucketed gradient exchange overlapped with backward, see the module docstring."""

    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph=True, process_group=None,
                 n_buckets=None, params=None, comm=None, storage=None, side_wgrad=None):
        self.model = model
        # channel counts that are not multiples of 4: the trainer trains the net's zero-padded TWIN (unet.PaddedTwin; padded entries have
        # exactly-zero gradients, so Adam never moves them) and cuts the parameters back into `model` at the host-visible points
        # (check_sync() / sync_to_module()); sync_from_module() re-embeds after `model`'s parameters were written from outside
        self._twin = None
        if getattr(model, "_n3d_padded", False):
            from . import unet as _unet
            tw = self._twin = model._n3d_make_twin()
            tw.embed(model)
            tw.twin.forward_loss = lambda x, t, smooth=1e-6, _tw=tw.twin: _unet.run_loss(_tw, x, t, None, smooth)
            model = tw.twin
        self.net = model      # the module whose kernels run (the model itself, or its padded twin)
        # side_wgrad (default: N3D_SIDE_WGRAD, on): the weight-gradient kernels of the C in {4, 8} levels -- nothing on the
        # backward chain waits for them -- are queued during the backward walk and launched on a SIDE HIP stream at a few cut
        # points (cell boundaries), as separately launched graphs tied to the main chain by events; the streams join once,
        # in front of the slab reduction + Adam.  (Round 3: two per-SAMPLE chains measured no gain -- batching already is that
        # overlap -- while the step's DAG has this slack: tools/two_chain_probe.py, tools/seg_overlap.cpp.)
        env = os.environ.get("N3D_SIDE_WGRAD", "1")
        self.side_wgrad = (env != "0") if side_wgrad is None else bool(side_wgrad)
        # graph mode captures BOTH schedules, times them on the real step and replays the faster one ("force": no comparison);
        # a replayed step is re-timed every 256 steps and the trainer falls back to the plain graph if the side schedule degrades
        # (e.g. another library created hardware queues and the two streams are time-sliced)
        self._side_force = side_wgrad == "force" or (side_wgrad is None and env == "force")
        self._side_explicit = side_wgrad is not None     # an eager trainer (graph=False) takes the side schedule only when asked to
        self._use_side = False
        self._capturing_side = False
        self._side_wsegs = 1
        self._side_retired = False   # recover() after a timed-out hand-off: the rest of the run stays on one stream
        self._n_steps = 0            # step() calls, on every path (the replay monitor's collective verdict is placed by this count)
        self._bracket_now = False
        self.schedule_times = None   # (plain seconds per step, side seconds per step) measured at capture
        if storage is not None:
            from . import unet as _unet
            _unet.set_storage(model, storage)   # "bf16": bf16 activation storage on the HBM-bound levels (BASELINE configs[4])
        self.loss_fn = WeightedDiceLoss()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(model.parameters()).device
        reserve_side_streams(self.device)     # first thing on the GPU (see there), also when THIS trainer will not use them
        plist = list(params) if params is not None else list(model.parameters())
        self.fp = FlatParams(plist, self.device)
        self._pad_mask = None
        if self._twin is not None:
            # gradients of the twin's PADDED parameter entries are not zero (GroupNorm couples a padded channel to its group) and Adam would
            # move them by lr per step: they are masked in front of every update, so the padded entries stay exactly 0
            self._pad_mask = _pad_mask_for(self._twin, self.model, self.fp, self.device)
        self.use_graph = graph
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        # N3D_FORCE_DP=1: take the multi-GPU code path (all-reduce + eager Adam after the graph) even in a 1-rank group --
        # lets a single-GPU box exercise exactly what N > 1 runs
        self.dp_path = self.world > 1 or (dist.is_initialized() and os.environ.get("N3D_FORCE_DP") == "1")
        if n_buckets is None:
            n_buckets = int(os.environ.get("N3D_DP_BUCKETS", "1"))
        self.n_buckets = max(1, n_buckets)
        self._graph = None
        self._side_graphs = None
        self._static_x = self._static_t = self._static_loss = None
        self.ctx = K.StepContext(self.device)  # batched weight packing + deferred wgrad reductions
        self.lr_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)  # read by the Adam kernel
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.scheduler = _plateau_for(self, "lr", "set_lr")  # train.py:50: ReduceLROnPlateau(factor=0.5)
        # (a weight-gradient stream of its own lets the side stream run data gradients of the C <= 8 cells inline: fused.SIDE_PAIRS_BWD)
        self.side = SideSchedule(self.device, self.ctx, wgrad_stream=_fused.SIDE_PAIRS_BWD) if (self.side_wgrad and self.device.type == "cuda") else None
        if self.side is not None and self.side.stream is None:
            self.side = None
        self.sync = GradSync(self.fp.grad, self.pg, 1, None, comm, header=self.fp.grad_full)
        # data parallel: the side-stream probe is rank-local and timing-based, but every rank must take the same schedule decisions
        # (they decide which graphs exist and in which order the collectives go out): the side schedule is on only if EVERY rank has one
        if not self.sync.all_true(self.side is not None):
            self.side = None
        # the bucketed, overlapped exchange rides on the side-stream schedule (a closed bucket is reduced and sent from the
        # weight-gradient stream); without one the exchange is a single bucket behind the step
        self._buckets = self._bucket_plan() if (self.dp_path and self.n_buckets > 1 and self.side is not None) else None
        if self._buckets is not None:
            self.side.tail_inline = ()      # every weight-gradient group on ONE stream: a closed bucket's slabs are reduced there
            self.sync.ranges = [r for _, r in self._buckets]
        self.sync.broadcast(self.fp.flat)

    # -- bucket plan ------------------------------------------------------------------------------
    def _bucket_plan(self):
        """[(cell index whose backward completes the bucket (-1 = the stems, i.e. the end), (begin, end) flat range)] in issue
        order, or None when the model is not a stems / down_cells / up_cells / last_conv net with a fusable head.
        Backward finishes the head first, then the cells last to first, then the stems, and `model.parameters()` lists
        stems, cells, head -- so what is complete after cell k is the TAIL of the flat buffer starting at cell k's first
        parameter.  Bucket j closes at the first point of the backward walk where at least thresholds[j] of the gradient
        bytes are complete; the last bucket closes at the end."""
        m = self.net
        if not all(hasattr(m, a) for a in ("stem0", "stem1", "down_cells", "up_cells", "last_conv")):
            return None
        off = {id(p): (o, o + (p.numel() + 3) // 4 * 4) for p, o in zip(self.fp.params, self.fp.offsets)}
        cells = list(m.down_cells) + list(m.up_cells)
        order = [m.stem0, m.stem1] + cells + [m.last_conv]
        spans, pos = [], 0
        for mod in order:
            ps = list(mod.parameters())
            if not ps or any(id(q) not in off for q in ps):
                return None
            lo, hi = min(off[id(q)][0] for q in ps), max(off[id(q)][1] for q in ps)
            if lo != pos:
                return None          # not the contiguous stems / cells / head order: keep the single bucket
            spans.append((lo, hi))
            pos = hi
        total = self.fp.numel
        if pos != total:
            return None
        n = self.n_buckets
        thresholds = [0.85] if n == 2 else [0.4 + (0.92 - 0.4) * j / (n - 2) for j in range(n - 1)]
        plan, end, j = [], total, 0
        for k in reversed(range(len(cells))):
            if j >= len(thresholds):
                break
            start = spans[2 + k][0]
            if (total - start) / total >= thresholds[j] and start < end and k > 0:
                plan.append((k, (start, end)))
                end = start
                j += 1
        plan.append((-1, (0, end)))
        return plan if len(plan) > 1 else None

    # -- pieces ---------------------------------------------------------------------------------
    def _fwd_bwd(self, x, t):
        with K.step_context(self.ctx):
            self.ctx.pack_all()            # one launch packs every conv weight for this step
            loss = _loss_of(self.net, self.loss_fn, x, t)
            prev, _fused.REUSE_GRAD_OUTPUT = _fused.REUSE_GRAD_OUTPUT, True   # this backward is all ours (no hooks, no retain)
            try:
                loss.backward(self._one)  # seed gradient kept resident: no fill launch per step
            finally:
                _fused.REUSE_GRAD_OUTPUT = prev
            self.ctx.flush_final()         # one launch finishes every weight-gradient reduction
        if not self.ctx.frozen:
            self.ctx.freeze()              # first pass only recorded which weights / layouts are needed
        return loss.detach()

    def _pipeline(self, x, t, cell_hook):
        """forward + Dice + backward WITHOUT autograd: fused.NetFn and head.HeadDiceFn are called directly, so the backward walk
        is one Python function on this thread and can hand over work on the way: cell_hook(k) is called when the backward of cell k
        (-1: the stems, i.e. the end) has been launched."""
        from . import head as _head, programs as _P
        m = self.net
        plan = getattr(m, "_net_plan", None)
        if not _fused.current(plan):
            plan = m._net_plan = _fused.net_plan(m, supernet=False)
        op = m.last_conv[0]
        hook = cell_hook
        with torch.no_grad(), K.step_context(self.ctx):
            self.ctx.pack_all()
            nctx = _Ctx((False, False) + (False,) * 4 + (True,) * len(plan.params))
            prev_pl, _fused.PLANAR_OUT = _fused.PLANAR_OUT, _fused.PLANAR_LAST     # (_direct_ok: the head is the fused one)
            try:
                body = _fused.NetFn.forward(nctx, plan, x, None, None, None, None, *plan.params)
            finally:
                _fused.PLANAR_OUT = prev_pl
            gate = _P.draw_gate(op.dropout, op.training, *_head.feat_shape(body), body.device)
            hctx = _Ctx((False, False, True, False, True, True))
            hctx.skip_p = True        # the step returns the loss only: the head does not write the probabilities
            loss, _ = _head.HeadDiceFn.forward(hctx, gate, float(self.loss_fn.smooth), body, t, op.conv.weight, op.conv.bias)
            dbody = _head.HeadDiceFn.backward(hctx, self._one, None)[2]
            prev, _fused.REUSE_GRAD_OUTPUT = _fused.REUSE_GRAD_OUTPUT, True
            prev_hook, _fused.CELL_DONE_HOOK = _fused.CELL_DONE_HOOK, hook
            try:
                _fused.NetFn.backward(nctx, dbody)
            finally:
                _fused.REUSE_GRAD_OUTPUT, _fused.CELL_DONE_HOOK = prev, prev_hook
        if not self.ctx.frozen:
            self.ctx.freeze()
        return loss

    def _direct_ok(self):
        """can the step run as the autograd-free pipeline (a stems / cells / fusable-head net with the Dice loss)?"""
        from . import head as _head
        m = self.net
        return (_fused.WHOLE_NET and isinstance(self.loss_fn, WeightedDiceLoss) and not hasattr(m, "kernel")
                and all(hasattr(m, a) for a in ("stem0", "stem1", "down_cells", "up_cells", "last_conv"))
                and _head.fusable(m.last_conv, torch.empty((1, m.last_conv[0].conv.weight.shape[1], 1, 1, 1), device="meta")))

    def _side_ok(self):
        return self.side is not None and not getattr(self, "_side_retired", False) and self._direct_ok()

    def check_sync(self):
        """synchronising check (call it at every host-visible point: before a checkpoint is written, at the end of an epoch, before
        a loss is reported): raises if a device-side wait of the side-stream schedule has timed out.  The updates of such steps
        were withheld on the device, so the weights are those of the last good step; `recover()` continues on one stream."""
        if self.side is not None:
            self.side.check()
        self.sync_to_module()

    def sync_to_module(self):
        """padded twin: the trained parameters back into the user's module (reference shapes); nothing to do otherwise"""
        if self._twin is not None:
            tp = dict(self._twin.twin.named_parameters())
            with torch.no_grad():
                for n, r in self.model.named_parameters():
                    r.copy_(self._twin.extract(n, tp[n].detach()))

    def sync_from_module(self):
        """padded twin: the user's module was written from outside (load_state_dict): embed its parameters again"""
        if self._twin is not None:
            self._twin.embed(self.model)

    def _padctx(self):
        """while kernels of a padded twin are being launched: conv-bias gradients by summation, per-term GroupNorm launches (the node-level
        ones share their element count with SE gates), no node-planar inner cells (unet.run_padded does the same)"""
        import contextlib
        return _padded_flags() if self._twin is not None else contextlib.nullcontext()

    def sync_timeouts(self):
        """device-side waits that have timed out since the trainer was built (synchronises; 0 without a side schedule)"""
        return int(self.side.sync[1].item()) if self.side is not None else 0

    def _poll(self):
        if self.side is not None:
            self.side.poll()

    def recover(self):
        """after a time-out (check_sync / step raised): acknowledge it and go on WITHOUT the side-stream schedule.  Nothing has to
        be restored -- the guarded updates of the affected steps never ran -- but those batches are lost to training."""
        if self.side is None:
            return
        self.side.acknowledge()
        self._use_side = False
        self._retire_side_graphs()
        self._side_retired = True
        if self._graph is None:
            self._static_x = self._static_t = None      # "force" had no plain graph: the next step captures one

    def _retire_side_graphs(self):
        g, self._side_graphs = self._side_graphs, None
        if g is not None:
            SideSchedule.raw_destroy(g[1])

    def close(self):
        """release the raw HIP graphs of the side streams (torch's own graphs and pools go with the object)"""
        try:
            self._retire_side_graphs()
        except Exception:
            pass

    def __del__(self):
        self.close()

    def _guard(self, loss=None):
        """UpdateGuard of this trainer's optimizer launches (None without a side schedule: nothing can time out)"""
        if self.side is None:
            return None
        return self.side.guard(loss, self.fp.grad_full.data_ptr() if self.dp_path else None)

    def _side_pass(self, x, t):
        """the autograd-free pipeline on two streams (SideSchedule): every deferrable weight-gradient launch is queued and handed to
        the side stream at the cuts of the backward walk (live), and the off-chain pieces of the net itself run there inline
        (fused.SIDE_FWD / SIDE_BWD).  Forward and backward run on this thread, so one arena covers both.  Returns the loss."""
        sd = self.side
        sd.begin_pass()
        with sd.forward_mode(), sd.backward_mode(), sd.deferring():
            cut_hook = _fused.CELL_DONE_HOOK          # (the deferring's: a cut of the weight-gradient queue at every cell boundary)
            if self._buckets is None:
                return self._pipeline(x, t, cell_hook=cut_hook)
            # bucketed exchange (data parallel) on device flags: a bucket that the backward walk has completed is closed on the
            # weight-gradient stream (SideSchedule.close_bucket) and all-reduced THERE, under the rest of the backward; the last
            # bucket (the head of the flat buffer, with the hand-off flag in front) goes out behind the tail, on the step's stream
            closes = {k: j for j, (k, _) in enumerate(self._buckets[:-1])}

            def hook(k):
                cut_hook(k)
                j = closes.get(k)
                if j is not None:
                    sd.close_bucket(lambda: self._bucket_closed(j))
            return self._pipeline(x, t, cell_hook=hook)

    def _bucket_closed(self, j):
        """bucket j is complete on the weight-gradient stream: while the streams are being captured the stream's graph is cut here
        (the all-reduce is launched between the segments at replay), eagerly the all-reduce goes out now"""
        sd = self.side
        if self._capturing_side:
            sd.split_wstream_capture()
        else:
            with torch.cuda.stream(sd.wstream):
                self.sync.reduce_range(j)

    def _side_step_eager(self, x, t):
        loss = self._side_pass(x, t)
        self.side.launch_side(redirect=True)
        self.side.finish()
        return loss

    def _allreduce_last(self):
        """bucketed exchange on flags: the buckets closed during the backward walk are on their way (weight-gradient stream, joined
        by the tail); the last one -- with the hand-off flag in front of it -- goes out here, on the step's stream"""
        if self.sync.active:
            K.guard_flag(self.side.ptr(1), self.side.ptr(4), self.fp.grad_full.data_ptr())
            self.sync.reduce_range(len(self.sync.ranges) - 1)

    def _allreduce(self):
        # word 0 of the gradient header: "a hand-off of THIS rank timed out" -- summed over the ranks with the gradients, so that
        # every rank withholds the same update (a rank without a side schedule contributes 0)
        if self.sync.active:
            if self.side is not None:
                K.guard_flag(self.side.ptr(1), self.side.ptr(4), self.fp.grad_full.data_ptr())
            else:
                self.fp.grad_full[:1].zero_()
        self.sync.all_reduce()

    def _update(self, loss=None):
        if self._pad_mask is not None:
            self.fp.grad.mul_(self._pad_mask)
        self.fp.adam(self.lr, self.betas, self.eps, 0.0, 1.0 / self.world, self.lr_dev, self._guard(loss))

    def set_lr(self, lr):
        """new learning rate for the following steps (also inside an already captured graph)"""
        self.lr = float(lr)
        self.lr_dev.fill_(self.lr)

    def _eager(self, x, t, allow_side=True):
        """allow_side=False: single stream whatever the trainer was built with.  The side-stream schedule ties its streams together
        with device-side waits, and anything that makes the HOST wait for the device in the middle of a pass (the caching allocator
        returning memory to the driver when a new shape does not fit its cache, for one) leaves those waits spinning until their
        time-out; a replayed graph allocates nothing, an eager pass on a shape seen for the first time may."""
        if allow_side and self._side_ok() and (self._side_explicit or self.use_graph):
            loss = self._side_step_eager(x, t)
            if self.dp_path:
                if self._buckets is not None:
                    self._allreduce_last()
                else:
                    self._allreduce()
        else:
            loss = self._fwd_bwd(x, t)
            if self.dp_path:
                self._allreduce()
        self._update(loss)
        return loss

    # -- public ---------------------------------------------------------------------------------
    def step(self, x, t):
        with self._padctx():
            return self._step(x, t)

    def _step(self, x, t):
        self._poll()     # host-only: a withheld update (timed-out hand-off) is fatal until recover()
        self._n_steps += 1
        if not self.use_graph:
            return self._eager(x, t)
        self._side_verdict()
        if self._static_x is None or (self._graph is None and self._side_graphs is None):
            self._capture(x, t)
        if x.shape != self._static_x.shape or t.shape != self._static_t.shape:
            # a batch of another shape (the reference's generator yields a smaller last batch of an epoch): the captured
            # graph is for one shape only, so this step runs eagerly (same kernels, same update) -- on ONE stream (see _eager)
            return self._eager(x, t, allow_side=False)
        # a caller that fills the trainer's own input buffers (input_buffers(): the data step writes the batch straight into them)
        # has no copy to pay; any other tensor is copied in
        if x is not self._static_x:
            self._static_x.copy_(x)
        if t is not self._static_t:
            self._static_t.copy_(t)
        if self._use_side:
            self._watched_side_replay()
            return self._side_loss
        self._graph.replay()
        if self.dp_path:
            self._allreduce()
            self._update(self._static_loss)
        return self._static_loss

    def input_buffers(self):
        """(x, t) device buffers the captured graphs read (None before the first graph step).  A data step that writes a batch
        straight into them (datastep.patch_batch(out=...)) and passes THESE tensors to step() skips the per-step input copy."""
        return self._static_x, self._static_t

    def _replay_side(self, exchange=True):
        # three graphs, no host-side cross-stream dependency: the streams meet through device flags (SideSchedule)
        # (the side graph goes first: its device-side waits are then in place when the main chain reaches its cuts, also when the
        # host is slower at launching than the GPU at running, e.g. under a profiler)
        g_main, side_exec, g_tail = self._side_graphs
        nseg = self._side_wsegs
        if nseg <= 1:
            self.side.raw_replay(side_exec)
            g_main.replay()
            g_tail.replay()
            if self.dp_path and exchange:
                self._allreduce()
                self._update(self._side_loss)
            return
        # bucketed exchange on flags: the weight-gradient stream's graph comes in segments, bucket j's all-reduce between segment j
        # and j + 1 on that stream (no host-side dependency: the segments wait for the chain's flags on the device)
        head, wsegs = side_exec[:-nseg], side_exec[-nseg:]
        self.side.raw_replay(head + wsegs[:1])
        g_main.replay()
        for j in range(1, nseg):
            if exchange:
                with torch.cuda.stream(self.side.wstream):
                    self.sync.reduce_range(j - 1)
            self.side.raw_replay(wsegs[j:j + 1])
        g_tail.replay()
        if exchange:
            self._allreduce_last()
            self._update(self._side_loss)

    def _replay_plain(self, exchange=True):
        self._graph.replay()
        if self.dp_path and exchange:
            self._allreduce()
            self._update(self._static_loss)

    def _watched_side_replay(self):
        """one replayed step of the side schedule.  Every 256th STEP (counted by step() on every path: a rank that ran an eager step
        for an odd-shaped last batch, or recaptured, still counts it) is bracketed with events; the verdict on that sample is taken
        at the start of the next step (_side_verdict)."""
        sd = self.side
        sd.replays += 1
        if self._bracket_now:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            self._replay_side()
            e1.record()
            sd.watch = (e0, e1)
            return
        self._replay_side()

    def _side_verdict(self):
        """start of a step, one step after a bracketed one: the trainer checks the time-outs of the device-side waits (raises) and the
        bracketed step's GPU time against the plain schedule's measured time -- a side schedule that has become slower than the plain
        graph is dropped.  Data parallel: the verdict is COLLECTIVE, and its place in the stream of collectives is a function of the
        STEP COUNT alone (every rank calls step() the same number of times, whatever path each step takes on a rank; a rank without
        a sample -- its bracketed step ran eagerly -- votes "fast"): one tiny all-reduce of "my sample was fast", issued before
        anything of this step.  A rank that retired its side schedule on its own would go on with another fp32 summation order than
        its peers and the replicas would drift apart bit by bit."""
        sd = self.side
        n = self._n_steps
        due, self._bracket_now = self._bracket_now, False
        # is THIS step a bracketed one?  (uniform over the ranks: the step count, or a re-check every rank agreed on one step ago)
        want = sd is not None and self._use_side and (n % 256 == 0 or getattr(sd, "recheck", 0) > 0)
        if want:
            sd.recheck = 0
        if sd is not None and self._use_side and due:
            fast, ms = True, None
            if sd.watch is not None:
                e0, e1 = sd.watch
                sd.watch = None
                e1.synchronize()
                sd.check()      # (the per-step poll() has seen a withheld update long before; this also catches a time-out whose
                                # step was not followed by an update yet)
                ms = e0.elapsed_time(e1)
                fast = self.schedule_times is None or ms * 1e-3 <= 1.5 * self.schedule_times[0] + 2e-4
            slow = not (self.sync.all_true(fast) if self.dp_path else fast)
            # ONE slow sample proves nothing -- the bracket also holds whatever the host did between the three graph launches (a
            # 100 000-step soak dropped a healthy schedule on a single 3.5 ms sample): a slow sample is measured again, three in a row
            # retire the schedule
            sd.slow_run = (getattr(sd, "slow_run", 0) + 1) if slow else 0
            sd.recheck = 1 if (slow and sd.slow_run < 3) else 0
            if sd.slow_run >= 3:
                import warnings
                warnings.warn("nas_3d_unet_amd: the side-stream schedule degraded (%s ms per step on this rank against %.2f ms for the plain "
                              "graph, three samples in a row%s); falling back to the plain graph"
                              % ("%.2f" % ms if ms is not None else "?", self.schedule_times[0] * 1e3, " on at least one rank" if self.dp_path else ""))
                self._use_side = False
                if self._graph is None:      # ("force": no plain graph was captured -- the next step captures one)
                    torch.cuda.synchronize(self.device)
                    self._retire_side_graphs()
                    self._side_retired = True
                    self._static_x = self._static_t = None
        self._bracket_now = bool(want and self._use_side)

    def _choose_schedule(self):
        """time the two captured schedules on the real step (state saved and restored around it) and keep the faster one"""
        if self._side_force:
            self._use_side = True
            return
        snap = _Snapshot([self.fp.flat, self.fp.exp_avg, self.fp.exp_avg_sq, self.fp.step, self.fp.grad_full] + _dropout_states(self.net))
        # data parallel: the timing runs replay the graphs only -- no collective, no update -- so a rank whose timing differs (or
        # whose side streams misbehave) cannot mispair all-reduces with its peers; the DECISION is then agreed on (MIN over ranks)
        tp = _time_schedule(lambda: self._replay_plain(exchange=False), self.device)
        ts = _time_schedule(lambda: self._replay_side(exchange=False), self.device)
        bad = self.side.sync_timeouts_now() != self.side.seen
        snap.restore()
        if bad:
            # hand-offs timed out while the side schedule was being timed: the state is restored, the schedule is not used
            self.side.acknowledge()
        torch.cuda.synchronize(self.device)
        self.schedule_times = (tp, ts)
        self._use_side = self.sync.all_true(ts < tp and not bad)
        if not self._use_side:
            self._retire_side_graphs()     # the plain graph won: the side graphs (and their raw HIP executables) are released

    def _capture(self, x, t):
        """capture (and, with a side stream, choose the schedule); the Dropout3d generators come out as they went in, so the
        masks of the training run do not depend on how many warm-up / timing passes the capture needed"""
        snap = _dropout_snapshot(self.net, self.device)
        try:
            self._capture_impl(x, t)
        finally:
            snap.restore()

    def _capture_impl(self, x, t):
        self._static_x = x.clone()
        self._static_t = t.clone()
        sided = self._side_ok()
        # warm-up on a side stream (allocator + lazy module state); no optimizer launch, the weights stay as they are
        s = capture_stream(self.device)
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(2):
                if sided:
                    self._side_step_eager(self._static_x, self._static_t)
                if not (sided and self._side_force):
                    self._fwd_bwd(self._static_x, self._static_t)
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if sided:
            self._capture_side(s)
            if self._side_force:
                self._use_side = True
                return
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, capture_error_mode="thread_local"):  # RCCL's watchdog thread may touch the runtime meanwhile
            self._static_loss = self._fwd_bwd(self._static_x, self._static_t)
            if not self.dp_path:
                self._update(self._static_loss)
        self._graph = g
        if sided:
            self._choose_schedule()

    def _capture_side(self, s):
        """The side-stream schedule as three HIP graphs: the main chain (torch capture: forward, Dice, backward with a flag store at
        every hand-off), the side stream's work behind its device-side waits -- captured AT THE SAME TIME as a raw HIP graph, since
        its launches are issued in the middle of the main chain's -- and the tail (wait for the side stream's 'done' flag, slab
        reduction, Adam).  The tail is a graph of its own because the slab-reduction job table is complete only once every
        weight-gradient launch has been issued."""
        import gc
        gc.collect()
        pool = torch.cuda.graph_pool_handle()
        sd = self.side
        g_main, g_tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        torch.cuda.synchronize()
        sd.raw_capture_begin()
        self._capturing_side = True
        try:
            with torch.cuda.stream(s):
                g_main.capture_begin(pool=pool, capture_error_mode="thread_local")
                self._side_loss = self._side_pass(self._static_x, self._static_t)
                g_main.capture_end()
            sd.launch_side(redirect=True)
        finally:
            self._capturing_side = False
            side_exec = sd.raw_capture_end()
        with torch.cuda.stream(s):
            g_tail.capture_begin(pool=pool, capture_error_mode="thread_local")
            sd.finish()
            if not self.dp_path:
                self._update(self._side_loss)
            g_tail.capture_end()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._side_graphs = (g_main, side_exec, g_tail)
        self._side_wsegs = sd.wseg_count

class SearchTrainer:
    """Supernet search step, first-order DARTS as in the reference (search.py:211-238):
    architecture pass on the validation batch (Adam on the four alpha matrices), then weight pass on the
    training batch (Adam on the kernel weights); both Adam instances use torch defaults (search.py:103-104).

    Exact-equivalent savings over the reference: the architecture pass does not compute weight gradients
    (the reference computes and discards them, search.py:231) and the weight pass does not compute alpha
    gradients -- requires_grad is switched per pass, so the corresponding kernels are simply not launched."""

    def __init__(self, shell, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, graph=True, process_group=None, comm=None, side_wgrad=None):
        self.model = shell
        # a kernel net with channel counts that are not multiples of 4: the trainer trains its zero-padded twin (see Trainer)
        self._twin = None
        self.net = shell
        if getattr(getattr(shell, "kernel", None), "_n3d_padded", False):
            tw = self._twin = shell.kernel._n3d_make_twin()
            tw.embed(shell.kernel)
            self.net = _TwinShell(shell, tw.twin)
        # the weight pass queues its weight-gradient kernels for the side stream (SideSchedule); the architecture pass has none
        env = os.environ.get("N3D_SIDE_WGRAD", "1")
        self.side_wgrad = (env != "0") if side_wgrad is None else bool(side_wgrad)
        self._side_force = side_wgrad == "force" or (side_wgrad is None and env == "force")   # as in Trainer: no comparison
        self._side_active = True     # _pass: use the side stream (when there is one)
        self.side_forward = True    # ... also for the off-chain edges of the forward passes
        self.side_backward = True   # ... and for the preprocess-fed edges of the backward passes
        # which preprocess-fed edges the side stream takes in the backward: the second input's only (round 4; "01" = both until then --
        # since the node levels' phases are single launches the chain absorbs the first input's terms in its own, wider, launches
        # and the side stream's last node level -- which the chain waits for at the end of every cell -- is half as long:
        # architecture pass 5.45 -> 5.25 ms, tools/search_phases.py)
        self.side_backward_inputs = (1,)
        self.side_backward_weight = ()      # (weight pass without a weight-gradient stream of its own: none)
        self._use_side = False
        self.schedule_times = None
        self.loss_fn = WeightedDiceLoss()
        self.lr, self.betas, self.eps = lr, betas, eps
        self.device = next(shell.parameters()).device
        reserve_side_streams(self.device)
        self.kparams = list(shell.kernel.parameters()) if self._twin is None else list(self._twin.twin.parameters())
        self.aparams = list(shell.alphas())
        self.fp = FlatParams(self.kparams, self.device)
        self._pad_mask = _pad_mask_for(self._twin, shell.kernel, self.fp, self.device) if self._twin is not None else None
        self.aflat, self.agrad, aoffs = flatten_params(self.aparams, self.device)
        self.a_m = torch.zeros_like(self.aflat)
        self.a_v = torch.zeros_like(self.aflat)
        self.a_step = K.step_counter(self.device) if self.device.type == "cuda" else torch.zeros(1, dtype=torch.int32, device=self.device)
        # the alphas' Adam state in the shape checkpoint.adam_state_dict reads (optim_shell, search.py:103)
        self.afp = types.SimpleNamespace(params=self.aparams, offsets=aoffs, exp_avg=self.a_m, exp_avg_sq=self.a_v, step=self.a_step)
        self._one = torch.ones((), dtype=torch.float32, device=self.device)
        self.ctx = K.StepContext(self.device)
        self.side = SideSchedule(self.device, self.ctx, wgrad_stream=True) if (self.side_wgrad and self.device.type == "cuda") else None
        if self.side is not None:
            # the weight pass' weight-gradient stream (219 launches) ends ~0.5 ms behind the chain: every other group of the walk's last
            # eleven goes to the inline side stream instead, which has little else to do in the backward (weight pass 5.57 + 0.51 ->
            # 5.78 + 0.09 ms; more groups there slow the chain by more than they take off the tail: tools/search_phases.py sweeps,
            # DESIGN.md section 5)
            self.side.tail_inline = (1, 3, 5, 7, 9, 11)
        if self.side is not None and self.side.stream is None:
            self.side = None
        if self.side is not None:
            self.side.early_finalize = 6      # 384 slab jobs per weight pass: reducing most of them early shrinks the tail 0.23 -> 0.15 ms
        self.use_graph = graph
        self._graph = None
        self._side_graphs = None
        # two learning rates (search.py:103-106): alphas ("shell") and kernel weights, each on its own plateau schedule
        self.lr_shell, self.lr_kernel = float(lr), float(lr)
        self.lr_shell_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
        self.lr_kernel_dev = torch.full((1,), float(lr), dtype=torch.float32, device=self.device)
        self.shell_scheduler = _plateau_for(self, "lr_shell", "set_shell_lr")
        self.kernel_scheduler = _plateau_for(self, "lr_kernel", "set_kernel_lr")
        # data parallel: two exchanges per step -- the alpha gradients (180 floats) after the architecture pass,
        # the kernel-weight gradients (27.4 MB) after the weight pass; weights and alphas broadcast once from rank 0
        self.pg = process_group
        self.world = dist.get_world_size(process_group) if (process_group is not None or dist.is_initialized()) else 1
        self.dp_path = self.world > 1 or (dist.is_initialized() and os.environ.get("N3D_FORCE_DP") == "1")
        self.sync_alpha = GradSync(self.agrad, self.pg, 1, None, comm, header=self.agrad._n3d_full)
        self.sync_kernel = GradSync(self.fp.grad, self.pg, 1, None, comm, header=self.fp.grad_full)
        self._graphs = None
        if not self.sync_kernel.all_true(self.side is not None):     # every rank the same schedule (see Trainer)
            self.side = None
        self.sync_kernel.broadcast(self.fp.flat)
        self.sync_kernel.broadcast(self.aflat)

    def set_shell_lr(self, lr):
        self.lr_shell = float(lr)
        self.lr_shell_dev.fill_(self.lr_shell)

    def set_kernel_lr(self, lr):
        self.lr_kernel = float(lr)
        self.lr_kernel_dev.fill_(self.lr_kernel)

    def _pass(self, x, t, arch, update=True, pack=True, side="inline"):
        """pack=False: the conv weights were packed by the pass before and have not changed since (the weight pass of a step
        follows the architecture pass, which only moves the alphas).
        side (weight pass with a SideSchedule): "inline" = queue the weight-gradient launches, then issue them on the side stream
        and join (eager use); "main" = only the main-stream part (the graph capture issues launch_side / finish itself)."""
        for p in self.kparams:
            p.requires_grad_(not arch)
        for p in self.aparams:
            p.requires_grad_(arch)
        if arch:
            self.agrad.zero_()  # alpha gradients arrive through autograd accumulation (softmax backward)
        # with a side stream BOTH passes use it: the forward of either pass runs the off-chain edges of every supernet cell there
        # (fused._run_forward_side), the weight pass also queues its weight-gradient launches for it
        sided = self.side is not None and self._side_active
        from . import programs as _P
        _P.PASS_TAG = "arch" if arch else "weight"
        with K.step_context(self.ctx):
            if pack:
                self.ctx.pack_all()
            if sided and self.side_forward:
                with self.side.forward_mode():
                    loss = _loss_of(self.net, self.loss_fn, x, t)
            else:
                loss = _loss_of(self.net, self.loss_fn, x, t)
            prev, _fused.REUSE_GRAD_OUTPUT = _fused.REUSE_GRAD_OUTPUT, True
            try:
                import contextlib
                # architecture pass: the side stream takes both preprocess-fed edges of every node (6 of a cell's 9: 7.8 -> 6.9 ms);
                # weight pass: it already carries every weight gradient -- with both edge sets on top the pass takes 9.6 instead of
                # 8.1 ms, with one 9.1 -- so its backward stays on one stream
                # -- unless they run on a stream of their own (SideSchedule.split): then the weight pass does the same, 8.1 -> 7.1 ms
                bwd_inputs = self.side_backward_inputs if (arch or (sided and self.side.split)) else self.side_backward_weight
                with (self.side.backward_mode(bwd_inputs) if (sided and self.side_backward and bwd_inputs) else contextlib.nullcontext()):
                    if sided and not arch:
                        with self.side.deferring():
                            loss.backward(self._one)
                    else:
                        loss.backward(self._one)  # seed gradient kept resident: no fill launch per step
            finally:
                _fused.REUSE_GRAD_OUTPUT = prev
            if not sided:
                self.ctx.flush_final()
        if sided and side == "main":
            return loss.detach()
        if sided:
            self.side.launch_side(redirect=True)
            self.side.finish()
        if not self.ctx.frozen and not arch:
            self.ctx.freeze()
        if update:
            self._update(arch, loss.detach())
        return loss.detach()

    def check_sync(self):
        """synchronising check for timed-out hand-offs (see Trainer.check_sync)"""
        if self.side is not None:
            self.side.check()
        self.sync_to_module()

    def sync_timeouts(self):
        return int(self.side.sync[1].item()) if self.side is not None else 0

    def recover(self):
        """after a time-out: acknowledge it and go on without the side-stream schedule (see Trainer.recover)"""
        if self.side is None:
            return
        self.side.acknowledge()
        self._use_side = False
        self._retire_side_graphs()
        self._side_active = False
        self._side_retired = True
        if self._graph is None and self._graphs is None:
            self._sx = None      # "force" had no plain graphs: the next step captures them

    def _retire_side_graphs(self):
        gs, self._side_graphs = self._side_graphs, None
        for g in gs or ():
            SideSchedule.raw_destroy(g[1])

    def close(self):
        """release the raw HIP graphs of the side streams"""
        try:
            self._retire_side_graphs()
        except Exception:
            pass

    def __del__(self):
        self.close()

    def _update(self, arch, loss=None):
        """exchange (data parallel) + Adam of the pass that just ran; guarded: a timed-out hand-off (on any rank) withholds it"""
        sd = self.side
        gbuf = self.agrad._n3d_full if arch else self.fp.grad_full
        sync = self.sync_alpha if arch else self.sync_kernel
        if self.dp_path:
            if sync.active:
                if sd is not None:
                    K.guard_flag(sd.ptr(1), sd.ptr(4), gbuf.data_ptr())
                else:
                    gbuf[:1].zero_()
            sync.all_reduce()
        guard = sd.guard(loss, gbuf.data_ptr() if self.dp_path else None) if sd is not None else None
        if arch:
            K.adam_step(self.aflat, self.agrad, self.a_m, self.a_v, self.a_step, self.lr_shell, self.betas[0], self.betas[1], self.eps,
                        grad_scale=1.0 / self.world, lr_dev=self.lr_shell_dev, guard=guard)
        else:
            if self._pad_mask is not None:
                self.fp.grad.mul_(self._pad_mask)       # (padded twin: the padded entries' gradients are not zero -- see Trainer)
            self.fp.adam(self.lr_kernel, self.betas, self.eps, 0.0, 1.0 / self.world, self.lr_kernel_dev, guard)

    def _both(self, x, t, vx, vt, update=True):
        la = self._pass(vx, vt, True, update)
        lw = self._pass(x, t, False, update, pack=False)   # same weights as the architecture pass just packed
        return la, lw

    def _capture_side(self, s):
        """Per pass three graphs: the main chain (torch capture), the side stream's work -- the forward's off-chain edges, then
        (weight pass) the queued weight-gradient groups -- captured on the side stream AT THE SAME TIME as a raw HIP graph, and
        the tail (join, slab reduction, Adam unless an exchange sits in between)."""
        import gc
        gc.collect()
        pool = torch.cuda.graph_pool_handle()
        sd = self.side
        torch.cuda.synchronize()
        graphs, losses = [], []
        for arch, (bx, bt) in ((True, (self._svx, self._svt)), (False, (self._sx, self._st))):
            g_main, g_tail = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            sd.raw_capture_begin()
            try:
                with torch.cuda.stream(s):
                    g_main.capture_begin(pool=pool, capture_error_mode="thread_local")
                    losses.append(self._pass(bx, bt, arch, update=False, pack=arch, side="main"))
                    g_main.capture_end()
                sd.launch_side(redirect=True)
            finally:
                side_exec = sd.raw_capture_end()
            with torch.cuda.stream(s):
                g_tail.capture_begin(pool=pool, capture_error_mode="thread_local")
                sd.finish()
                if not self.dp_path:
                    self._update(arch, losses[-1])
                g_tail.capture_end()
            graphs.append((g_main, side_exec, g_tail))
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        self._side_losses, self._side_graphs = tuple(losses), graphs

    def step(self, x, t, val_x, val_t):
        """returns (architecture-pass loss, weight-pass loss) as device scalars"""
        if self._twin is not None:
            with _padded_flags():
                return self._step(x, t, val_x, val_t)
        return self._step(x, t, val_x, val_t)

    def sync_to_module(self):
        """padded twin: the trained kernel weights back into shell.kernel (reference shapes)"""
        if self._twin is not None:
            tp = dict(self._twin.twin.named_parameters())
            with torch.no_grad():
                for n, r in self.model.kernel.named_parameters():
                    r.copy_(self._twin.extract(n, tp[n].detach()))

    def sync_from_module(self):
        if self._twin is not None:
            self._twin.embed(self.model.kernel)

    def _step(self, x, t, val_x, val_t):
        if self.side is not None:
            self.side.poll()     # host-only: a withheld update (timed-out hand-off) is fatal until recover()
        if not self.use_graph:
            return self._both(x, t, val_x, val_t)
        if getattr(self, "_sx", None) is None or (self._graph is None and self._graphs is None and self._side_graphs is None):
            drop_snap = _dropout_snapshot(self.net, self.device)   # restored below: masks independent of the warm-up passes
            self._sx, self._st, self._svx, self._svt = x.clone(), t.clone(), val_x.clone(), val_t.clone()
            # warm-up on a side stream (allocator + lazy module state) WITHOUT the optimizer launches: weights, alphas,
            # Adam moments and step counters -- possibly just loaded from a checkpoint (search.py:108-127) -- stay untouched
            s = capture_stream(self.device)
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                for _ in range(2):
                    self._both(self._sx, self._st, self._svx, self._svt, update=False)
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            sided = self.side is not None and not getattr(self, "_side_retired", False)
            if sided:
                self._capture_side(s)
            if sided and self._side_force:
                self._use_side = True
            else:
                self._side_active = False
                if self.dp_path:
                    # one graph per pass (forward + backward only); the exchange and Adam of each pass run eagerly behind it
                    pool = torch.cuda.graph_pool_handle()
                    ga, gw = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                    with torch.cuda.graph(ga, pool=pool, capture_error_mode="thread_local"):
                        la = self._pass(self._svx, self._svt, True, update=False)
                    with torch.cuda.graph(gw, pool=pool, capture_error_mode="thread_local"):
                        lw = self._pass(self._sx, self._st, False, update=False, pack=False)
                    self._losses, self._graphs = (la, lw), (ga, gw)
                else:
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):  # RCCL's watchdog thread may touch the runtime meanwhile
                        self._losses = self._both(self._sx, self._st, self._svx, self._svt)
                    self._graph = g
                self._side_active = sided
                if sided:
                    self._choose_schedule()
            drop_snap.restore()
        if any(a.shape != b.shape for a, b in ((x, self._sx), (t, self._st), (val_x, self._svx), (val_t, self._svt))):
            # remainder batch of an epoch: eager step (the graph is for one shape), on ONE stream (Trainer._eager says why)
            was, self._side_active = self._side_active, False
            try:
                return self._both(x, t, val_x, val_t)
            finally:
                self._side_active = was
        for dst, src in ((self._sx, x), (self._st, t), (self._svx, val_x), (self._svt, val_t)):
            if src is not dst:      # input_buffers(): a caller that fills the trainer's own buffers has no copy to pay
                dst.copy_(src)
        if self._use_side:
            self._replay_side()
            sd = self.side
            sd.replays += 1
            if sd.replays % 256 == 0:
                sd.check()
            return self._side_losses
        self._replay_plain()
        return self._losses

    def input_buffers(self):
        """(x, t, val_x, val_t) device buffers the captured graphs read (see Trainer.input_buffers)"""
        return self._sx, self._st, self._svx, self._svt

    def _replay_side(self, exchange=True):
        for k, (arch, (g_main, side_exec, g_tail)) in enumerate(zip((True, False), self._side_graphs)):
            self.side.raw_replay(side_exec)      # first: see Trainer._replay_side
            g_main.replay()
            g_tail.replay()
            if self.dp_path and exchange:
                self._update(arch, self._side_losses[k])

    def _replay_plain(self, exchange=True):
        if self._graphs is not None:
            self._graphs[0].replay()
            if exchange:
                self._update(True, self._losses[0])
            self._graphs[1].replay()
            if exchange:
                self._update(False, self._losses[1])
            return
        self._graph.replay()

    def _choose_schedule(self):
        """time the two captured schedules on the real step (state saved and restored around it) and keep the faster one"""
        snap = _Snapshot([self.fp.flat, self.fp.exp_avg, self.fp.exp_avg_sq, self.fp.step, self.fp.grad_full, self.aflat, self.agrad._n3d_full,
                          self.a_m, self.a_v, self.a_step] + _dropout_states(self.net))
        # data parallel: graphs only, no collective and no update while timing; the decision is agreed on (Trainer._choose_schedule)
        ex = not self.dp_path
        tp = _time_schedule(lambda: self._replay_plain(exchange=ex), self.device, warm=1, reps=3)
        ts = _time_schedule(lambda: self._replay_side(exchange=ex), self.device, warm=1, reps=3)
        bad = self.side.sync_timeouts_now() != self.side.seen
        snap.restore()
        if bad:
            self.side.acknowledge()
        torch.cuda.synchronize(self.device)
        self.schedule_times = (tp, ts)
        self._use_side = self.sync_kernel.all_true(ts < tp and not bad)
        if not self._use_side:
            self._retire_side_graphs()

