"""Thin Python wrappers: torch tensors -> pitched-NDHWC views -> libn3d C ABI calls.

torch is used for device memory (caching allocator) and the current HIP stream only; every
arithmetic kernel on the path is a libn3d kernel.  Logical tensor shape is the reference's
(B, C, D, H, W) (prim_ops.py / cell.py callers); physical layout is NDHWC with a voxel pitch
`ld` so that channel slices of a wider buffer (torch.cat(dim=1), cell.py:82) need no copy.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

from . import _lib
from ._lib import ConvBwdCall, ConvFwdCall, DwJob, GnFwdTerm, GnBwdTerm, PlainCoefTerm, SeTerm, ACCUMULATE, POOL_MAX, PREPACKED, RELU, RELU_IN, ConvGeom, FinalJob, N3DError, PackJob, check

__all__ = ["View", "as_view", "empty_ndhwc", "stream_ptr", "conv_geom", "ptr"]


# Launch redirection (train.SideSchedule, round 3): inside `with on_side(stream)` every libn3d launch goes to the given HIP stream
# while torch's current stream -- and with it every allocation -- stays where it is.  The caller orders the two streams with
# sync_signal / sync_wait and keeps every tensor of the redirected launches alive until the streams have joined.
_redirect = None


def stream_ptr():
    if _redirect is not None:
        return C.c_void_p(_redirect)
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class on_side:
    def __init__(self, stream):
        self.ptr = stream if isinstance(stream, int) else stream.cuda_stream

    def __enter__(self):
        global _redirect
        self.prev, _redirect = _redirect, self.ptr
        return self

    def __exit__(self, *exc):
        global _redirect
        _redirect = self.prev
        return False


def _side_launch_ptr(tensors):
    """Stream for work that nothing downstream on the main stream waits for (weight gradients).
    With a StepContext that owns a side stream the launch goes there, ordered after everything enqueued so far
    on the main stream; the tensors it touches are kept alive until the context joins the streams.
    (Allocation always happens on the main stream; only the launch moves.)"""
    if _ctx is None or _ctx.side is None:
        return stream_ptr()
    main = torch.cuda.current_stream()
    _ctx.side.wait_stream(main)
    _ctx.keep.extend(tensors)
    _ctx.side_dirty = True
    return C.c_void_p(_ctx.side.cuda_stream)


def ptr(t):
    """device pointer of a tensor (or None)"""
    if t is None:
        return None
    return C.c_void_p(t.data_ptr())


# Storage type of NEW forward activations (conv outputs, node buffers): fp32, or bf16 inside `with storage(torch.bfloat16)` --
# the bf16 configuration (BASELINE configs[4]) switches it per cell (fused._Plan.dt).  Gradient buffers never consult it: they
# take the type of the forward tensor they belong to (`like`).
_act_dtype = torch.float32
# bf16 configuration, the cells that keep fp32 storage (16 .. 64 channels): their MFMA conv kernels round the operands of the matrix
# products to bf16 in registers (include/n3d.h, N3D_MM_BF16) -- set per cell next to the storage type (fused._Plan.mm_bf16)
_mm_bf16 = False


class storage:
    def __init__(self, dtype, mm_bf16=False):
        self.dtype, self.mm = dtype, bool(mm_bf16)

    def __enter__(self):
        global _act_dtype, _mm_bf16
        self.prev, _act_dtype = _act_dtype, self.dtype
        self.prev_mm, _mm_bf16 = _mm_bf16, self.mm
        return self

    def __exit__(self, *exc):
        global _act_dtype, _mm_bf16
        _act_dtype, _mm_bf16 = self.prev, self.prev_mm
        return False


def _mm(flags):
    """conv flags + N3D_MM_BF16 while a cell of the bf16 configuration with fp32 storage is running"""
    return flags | (_lib.MM_BF16 if _mm_bf16 else 0)


def empty_ndhwc(B, Cc, D, H, W, device, dtype=None):
    """Dense NDHWC storage presented with the reference's logical (B, C, D, H, W) shape (dtype None: the current storage type)."""
    dtype = dtype if dtype is not None else _act_dtype
    if dtype == torch.bfloat16:
        # 16 bytes of slack behind the tensor: the bf16 3x3x3 kernels fetch 16 bytes per voxel by LDS-DMA, i.e. 8 bytes
        # past a 4-channel voxel (conv_bf16.hip)
        n = B * D * H * W * Cc
        return torch.empty(n + 8, device=device, dtype=dtype)[:n].view(B, D, H, W, Cc).permute(0, 4, 1, 2, 3)
    return torch.empty((B, D, H, W, Cc), device=device, dtype=dtype).permute(0, 4, 1, 2, 3)


def zeros_ndhwc(B, Cc, D, H, W, device, dtype=None):
    dtype = dtype if dtype is not None else _act_dtype
    if dtype == torch.bfloat16:      # same 16 bytes of slack as empty_ndhwc
        n = B * D * H * W * Cc
        return torch.zeros(n + 8, device=device, dtype=dtype)[:n].view(B, D, H, W, Cc).permute(0, 4, 1, 2, 3)
    return torch.zeros((B, D, H, W, Cc), device=device, dtype=dtype).permute(0, 4, 1, 2, 3)


def _bf16_slack_ok(t, ld):
    """bf16 tensors whose voxel is 8 bytes (4 channels) are read 16 bytes at a time by the LDS-DMA tile fills (conv_bf16.hip), i.e.
    up to 8 bytes past the last element: the storage behind the tensor must hold them (include/n3d.h, N3D_SRC_BF16)"""
    if t.dtype != torch.bfloat16 or t.shape[1] != 4:
        return True
    B, Cc, D, H, W = t.shape
    last = t.storage_offset() + (B * D * H * W - 1) * ld + Cc        # one past the last element, in elements
    return t.untyped_storage().nbytes() >= (last + 4) * 2


def like(v):
    """fresh dense View of the shape and storage type of View v (gradient buffers, temporaries); a node-planar tensor gives a node-planar one"""
    if isinstance(v, Planar):
        return empty_planar(v.nn, v.B, v.cn, v.D, v.H, v.W, v.t.device, v.t.dtype)
    return View(empty_ndhwc(v.B, v.C, v.D, v.H, v.W, v.t.device, v.t.dtype), v.C)


def _need_f32(what, *views):
    for v in views:
        if v is not None and v.dt != _lib.F32:
            raise N3DError("%s: bf16 storage is not built for this kernel family (fp32 tensors only)" % what)


def _cflags(flags, src, dst):
    """conv-family storage flags: src / dst = the first / second activation tensor of the call"""
    return _mm(flags) | (_lib.SRC_BF16 if src.dt == _lib.BF16 else 0) | (_lib.DST_BF16 if dst.dt == _lib.BF16 else 0)


def _aflag(v):
    return _lib.ACT_BF16 if v.dt == _lib.BF16 else 0


def _same_dt(what, *views):
    dts = {v.dt for v in views if v is not None}
    if len(dts) > 1:
        raise N3DError("%s: the activation tensors of one epilogue call must share a storage type" % what)


class View:
    """Pitched NDHWC view of a logical (B, C, D, H, W) device tensor (fp32, or bf16 storage: dt = N3D_F32 / N3D_BF16;
    the pitch counts elements)."""
    __slots__ = ("t", "p", "ld", "B", "C", "D", "H", "W", "N", "dt")

    def __init__(self, t, ld):
        self.t = t
        self.p = C.c_void_p(t.data_ptr())
        self.ld = int(ld)
        self.dt = _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32
        self.B, self.C, self.D, self.H, self.W = (int(s) for s in t.shape)
        self.N = self.D * self.H * self.W


class Planar:
    """Node-planar feature map (include/n3d.h, n3d_head): the concatenation of a cell's `nn` node outputs kept as `nn` DENSE
    (B, cn, D, H, W) NDHWC tensors in one allocation -- storage (nn, B, D, H, W, cn), presented to torch as the 6-D tensor
    `t` of shape (nn, B, cn, D, H, W).  Readers: the fused head (n3d_head.node_c) and the 1x1x1 preprocess convs, which take it as ONE
    tensor whose pitch `ld` = cn is smaller than its channel count C = nn cn (include/n3d.h, "node-planar tensors"): `p` points to node 0.
    `nodes[k]` is the ordinary dense View of node k."""
    __slots__ = ("t", "nodes", "nn", "cn", "B", "C", "D", "H", "W", "N", "dt", "node_stride", "p", "ld")

    def __init__(self, t):
        self.t = t
        self.nn, self.B, self.cn, self.D, self.H, self.W = (int(v) for v in t.shape)
        self.C, self.N = self.nn * self.cn, self.D * self.H * self.W
        self.dt = _lib.BF16 if t.dtype == torch.bfloat16 else _lib.F32
        self.node_stride = self.B * self.N * self.cn
        self.nodes = [View(t[k], self.cn) for k in range(self.nn)]
        self.p, self.ld = self.nodes[0].p, self.cn


def empty_planar(nn, B, cn, D, H, W, device, dtype=None):
    dtype = dtype if dtype is not None else _act_dtype
    n = nn * B * D * H * W * cn
    slack = 8 if dtype == torch.bfloat16 else 0      # bf16: the 16-byte-per-voxel LDS-DMA reads run 8 bytes past a 4-channel voxel
    return Planar(torch.empty(n + slack, device=device, dtype=dtype)[:n].view(nn, B, D, H, W, cn).permute(0, 1, 5, 2, 3, 4))


def as_planar(t, what="tensor"):
    """a 6-D (nn, B, cn, D, H, W) tensor with dense (nn, B, D, H, W, cn) storage -> Planar (repacked with a copy otherwise)"""
    if not isinstance(t, torch.Tensor) or t.dim() != 6 or not t.is_cuda or t.dtype not in (torch.float32, torch.bfloat16):
        raise N3DError("%s: expected a node-planar (nn, B, cn, D, H, W) device tensor" % what)
    nn, B, cn, D, H, W = t.shape
    want = (B * D * H * W * cn, D * H * W * cn, 1, H * W * cn, W * cn, cn)
    if tuple(t.stride()) != want or t.data_ptr() % (4 * t.element_size()) != 0:
        n = empty_planar(nn, B, cn, D, H, W, t.device, t.dtype)
        n.t.copy_(t)
        return n
    return Planar(t)


def _pitch_of(t):
    """Return the voxel pitch if `t` is a pitched NDHWC view, else None."""
    B, Cc, D, H, W = t.shape
    s = t.stride()
    if W > 1:
        ld = s[4]
    elif H > 1:
        ld = s[3]
    elif D > 1:
        ld = s[2]
    elif B > 1:
        ld = s[0]
    else:
        ld = Cc
    if ld < Cc:
        return None
    if Cc > 1 and s[1] != 1:
        return None
    if W > 1 and s[4] != ld:
        return None
    if H > 1 and s[3] != W * ld:
        return None
    if D > 1 and s[2] != H * W * ld:
        return None
    if B > 1 and s[0] != D * H * W * ld:
        return None
    return ld


def as_view(t, what="tensor", bf16_ok=True):
    """Validate (or repack with a copy) a logical (B,C,D,H,W) tensor into a pitched NDHWC view (fp32 or bf16 storage; the
    kernel families without bf16 support check for themselves)."""
    if not isinstance(t, torch.Tensor) or t.dim() != 5:
        raise N3DError("%s: expected a 5-D (B,C,D,H,W) tensor" % what)
    if not t.is_cuda:
        raise N3DError("%s is on %s: the nas_3d_unet_amd ops only run on a HIP (gfx950) device; "
                       "there is no CPU fallback" % (what, t.device))
    if t.dtype != torch.float32 and not (bf16_ok and t.dtype == torch.bfloat16):
        raise N3DError("%s: fp32 expected, got %s" % (what, t.dtype))
    ld = _pitch_of(t)
    if ld is None or (t.data_ptr() % (4 * t.element_size()) != 0) or (ld % 4 != 0 and t.shape[1] % 4 == 0) or not _bf16_slack_ok(t, ld):
        B, Cc, D, H, W = t.shape
        n = empty_ndhwc(B, Cc, D, H, W, t.device, t.dtype)
        n.copy_(t)  # layout plumbing (strided copy); arithmetic stays in libn3d
        t, ld = n, Cc
    return View(t, ld)


def as_act(t, what="tensor"):
    """an activation (or its gradient) as the kernels take it: a 5-D tensor -> View, a 6-D node-planar one -> Planar"""
    if isinstance(t, torch.Tensor) and t.dim() == 6:
        return as_planar(t, what)
    return as_view(t, what)


def conv_geom(B, Di, Hi, Wi, Ci, Co, k, stride, dil, pad, depthwise=False):
    def od(i):
        return (i + 2 * pad - dil * (k - 1) - 1) // stride + 1
    return ConvGeom(B, Di, Hi, Wi, Ci, od(Di), od(Hi), od(Wi), Co, k, stride, dil, pad, 1 if depthwise else 0)


def _ws(g, device):
    n = _lib.load().n3d_conv_workspace_bytes(C.byref(g))
    return torch.empty(max(int(n), 256), dtype=torch.uint8, device=device), int(n)


class StepContext:
    """Per-trainer launch batching (no reference counterpart; removes ~225 tiny launches per step):
      * every conv weight of the net is packed into its kernel layout by ONE n3d_pack_batch launch at the
        start of a step (first pass records the jobs, later passes hand the packed slot to the conv call);
      * the fixed-order reductions of all weight-gradient partial slabs run as ONE n3d_wgrad_finalize_batch
        launch at the end of backward."""

    def __init__(self, device, side_stream=False):
        self.device = device
        # optional side stream for the weight-gradient kernels (nothing on the main chain waits for them until the end
        # of backward).  Measured on MI355X / ROCm 7.2: the 75 cross-stream graph edges cost more than the overlap
        # buys (6.5 ms vs 5.7 ms per step), so it is OFF by default.
        self.side = torch.cuda.Stream(device=device) if side_stream else None
        self.side_dirty = False
        self.pending = {}      # key -> (weight, Co, Ci, taps, data_grad, layout, cdp, floats)
        self.slots = {}        # key -> (ptr, bytes)
        self.frozen = False
        self.buf = None
        self.jobs = None
        self.final = []
        self.keep = []
        # deferred weight-gradient LAUNCHES (train.Trainer's side-stream schedule): with defer_wgrad the weight-gradient kernels
        # nothing on the backward chain waits for are queued here (operands kept alive) and launched by flush_wgrads() on
        # whatever stream is current then
        self.defer_wgrad = False
        self.wq = []
        self.hold = []         # operands of launched-but-possibly-still-running side work, released by the trainer after the join
        self.keep_jobs = []
        self.final_side = []   # slab-reduction jobs of the launches flush_wgrads issued (the side stream's own)
        self.final_inline = [] # ... of the groups that went to the OTHER side stream (reduced by the tail only: an early reduction on the
                               # weight-gradient stream is not ordered behind them)

    # ---- weight packing
    def slot(self, w, g, data_grad, flags):
        lay, cdp, fl = C.c_int32(), C.c_int32(), C.c_int64()
        check(_lib.load().n3d_conv_pack_info(C.byref(g), 1 if data_grad else 0, flags, C.byref(lay), C.byref(cdp), C.byref(fl)),
              "n3d_conv_pack_info")
        if lay.value < 0:
            return None
        key = (w.data_ptr(), bool(data_grad), lay.value, cdp.value)
        if key in self.slots:
            return self.slots[key]
        if not self.frozen:
            self.pending[key] = (w, g.Co, g.Ci, g.k ** 3, 1 if data_grad else 0, lay.value, cdp.value, fl.value)
        return None

    def freeze(self):
        total = sum((v[7] + 63) // 64 * 64 for v in self.pending.values())
        self.buf = torch.empty(max(total, 64), dtype=torch.float32, device=self.device)
        arr = (PackJob * max(len(self.pending), 1))()
        off = 0
        for i, (key, (w, Co, Ci, taps, dg, lay, cdp, fl)) in enumerate(self.pending.items()):
            dst = self.buf.data_ptr() + off * 4
            arr[i] = PackJob(w.data_ptr(), dst, Co, Ci, taps, dg, lay, cdp)
            self.slots[key] = (dst, fl * 4)
            off += (fl + 63) // 64 * 64
        self.jobs, self.njobs = arr, len(self.pending)
        self.frozen = True

    def pack_all(self):
        if self.frozen and self.njobs:
            check(_lib.load().n3d_pack_batch(self.jobs, self.njobs, stream_ptr()), "n3d_pack_batch")

    # ---- deferred weight-gradient reductions
    def join(self):
        if self.side is not None and self.side_dirty:
            torch.cuda.current_stream().wait_stream(self.side)
            self.side_dirty = False

    def mark(self, tag):
        """a cut point in the queue: flush_wgrads calls on_mark(tag) when it gets there (the trainer puts a device-side wait)"""
        self.wq.append(("mark", tag))

    def queued(self):
        """launches queued since the last mark"""
        n = 0
        for item in reversed(self.wq):
            if item[0] == "mark":
                break
            n += 1
        return n

    def flush_wgrads(self, on_mark=None, inline=False):
        """launch every queued weight-gradient kernel on the CURRENT stream, in queue order (the caller orders the stream behind
        the producers: on_mark(tag) at every mark).  inline: the stream is not the weight-gradient stream (see final_inline)"""
        q, self.wq = self.wq, []
        n = 0
        for item in q:
            if item[0] == "mark":
                if on_mark is not None:
                    on_mark(item[1])
                continue
            launch, job, ws, keep = item
            if _DROP_SIDE:      # timing probe only (tools/side_timeline.py --drop): the main chain without its weight gradients
                continue
            launch(stream_ptr())
            if job.nchunks > 0:
                (self.final_inline if inline else self.final_side).append(job)
            self.hold.append((ws, keep))
            n += 1
        return n

    def finalize_now(self, n_main):
        """side stream: reduce the slabs of every weight gradient this stream has launched so far (in order behind them) and of the
        first n_main jobs of the main chain (the caller knows they are complete: the side stream has passed a wait on a flag the
        main stream stored behind them); the operands stay held until flush_final"""
        jobs = self.final_side + self.final[:n_main]
        if jobs:
            arr = (FinalJob * len(jobs))(*jobs)
            check(_lib.load().n3d_wgrad_finalize_batch(arr, len(jobs), stream_ptr()), "n3d_wgrad_finalize_batch")
            self.keep_jobs.append(arr)
            self.final_side = []
            self.final = self.final[n_main:]

    def flush_final(self):
        if self.wq:
            self.flush_wgrads()
        self.join()
        jobs = self.final + self.final_side + self.final_inline
        if jobs:
            arr = (FinalJob * len(jobs))(*jobs)
            check(_lib.load().n3d_wgrad_finalize_batch(arr, len(jobs), stream_ptr()), "n3d_wgrad_finalize_batch")
        self.final, self.final_side, self.final_inline, self.keep, self.hold, self.keep_jobs = [], [], [], [], [], []


_ctx = None
_DROP_SIDE = False


class step_context:
    """with step_context(ctx): conv calls use ctx's packed-weight slots and defer their wgrad reductions."""

    def __init__(self, ctx):
        self.ctx = ctx

    def __enter__(self):
        global _ctx
        self.prev, _ctx = _ctx, self.ctx
        return self.ctx

    def __exit__(self, *exc):
        global _ctx
        _ctx = self.prev
        return False


def deferring():
    """is the side-stream schedule queueing weight-gradient launches right now?"""
    return _ctx is not None and _ctx.defer_wgrad


class no_defer:
    """with no_defer(): weight gradients launch in place (their dy operand is a buffer the backward chain goes on writing)"""

    def __enter__(self):
        self.ctx = _ctx
        self.prev = _ctx.defer_wgrad if _ctx is not None else False
        if _ctx is not None:
            _ctx.defer_wgrad = False
        return self

    def __exit__(self, *exc):
        if self.ctx is not None:
            self.ctx.defer_wgrad = self.prev
        return False


def _packed(w, g, data_grad, flags, device):
    """(ws tensor or None, ws ptr, ws bytes, flags) for a forward / data-gradient conv call."""
    if _ctx is not None and not getattr(w, "_n3d_nopack", False):  # temporaries (zero-padded weights) are packed by the call itself
        sl = _ctx.slot(w, g, data_grad, flags)
        if sl is not None:
            return None, C.c_void_p(sl[0]), sl[1], flags | PREPACKED
    ws, n = _ws(g, device)
    return ws, ptr(ws), n, flags


# ------------------------------------------------------------------------------------------ convs
def conv_stats_rows(g, transposed, flags=0, x=None, y=None):
    """x / y: the call's input / output views (their storage types select the kernel, hence the row count)"""
    if x is not None and y is not None:
        flags = _cflags(flags, x, y)
    return int(_lib.load().n3d_conv_stats_rows(C.byref(g), 1 if transposed else 0, flags))


def stats_rows(N, Cc):
    return int(_lib.load().n3d_stats_rows(N, Cc))


def conv_fwd(g, x: View, w, bias, y: View, flags=0, in_gate=None, stats=None, transposed=False):
    flags = _cflags(flags, x, y)
    ws, wsp, n, flags = _packed(w, g, transposed, flags, x.t.device)
    fn = _lib.load().n3d_convT_fwd if transposed else _lib.load().n3d_conv_fwd
    check(fn(C.byref(g), x.p, x.ld, ptr(w), ptr(bias), y.p, y.ld, flags, ptr(in_gate), ptr(stats), wsp, n,
             stream_ptr()), "n3d_convT_fwd" if transposed else "n3d_conv_fwd")


def conv_bwd_data(g, dy: View, w, dx: View, flags=0, relu_src: View | None = None, out_gate=None, transposed=False):
    flags = _cflags(flags, dy, dx)
    if relu_src is not None and relu_src.dt != dx.dt:
        raise N3DError("conv_bwd_data: the ReLU mask source must have the gradient's storage type")
    ws, wsp, n, flags = _packed(w, g, not transposed, flags, dy.t.device)
    lib = _lib.load()
    if transposed:
        if relu_src is not None or out_gate is not None:
            raise N3DError("convT_bwd_data: relu/gate epilogue not supported")
        check(lib.n3d_convT_bwd_data(C.byref(g), dy.p, dy.ld, ptr(w), dx.p, dx.ld, flags, wsp, n, stream_ptr()),
              "n3d_convT_bwd_data")
    else:
        check(lib.n3d_conv_bwd_data(C.byref(g), dy.p, dy.ld, ptr(w), dx.p, dx.ld, flags,
                                    relu_src.p if relu_src is not None else None,
                                    relu_src.ld if relu_src is not None else 0, ptr(out_gate), wsp, n,
                                    stream_ptr()), "n3d_conv_bwd_data")


def conv_bwd_weight(g, x: View, dy: View, dw, dbias, flags=0, in_gate=None, transposed=False, defer=True):
    """defer=False: dw is complete when the call returns to the stream (no batched reduction at the end of backward)"""
    flags = _cflags(flags, x, dy)
    ws, n = _ws(g, x.t.device)
    lib = _lib.load()
    job = FinalJob() if (_ctx is not None and defer) else None
    jp = C.byref(job) if job is not None else None
    if transposed and in_gate is not None:
        raise N3DError("convT_bwd_weight: gate not supported")
    if job is not None and _ctx.defer_wgrad:
        # queued: launched by StepContext.flush_wgrads() (side stream); x / dy / ws stay alive until the trainer's join
        if transposed:
            def launch(sp, g=g, x=x, dy=dy, dw=dw, dbias=dbias, flags=flags, ws=ws, n=n, jp=jp):
                check(lib.n3d_convT_bwd_weight(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(dw), ptr(dbias), flags, ptr(ws), n, jp, sp),
                      "n3d_convT_bwd_weight")
        else:
            def launch(sp, g=g, x=x, dy=dy, dw=dw, dbias=dbias, flags=flags, in_gate=in_gate, ws=ws, n=n, jp=jp):
                check(lib.n3d_conv_bwd_weight(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(dw), ptr(dbias), flags, ptr(in_gate), ptr(ws), n, jp, sp),
                      "n3d_conv_bwd_weight")
        _ctx.wq.append((launch, job, ws, (x.t, dy.t, in_gate, dw, dbias)))
        return
    sp = _side_launch_ptr([x.t, dy.t, ws, in_gate]) if job is not None else stream_ptr()
    if transposed:
        check(lib.n3d_convT_bwd_weight(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(dw), ptr(dbias), flags, ptr(ws), n, jp,
                                       sp), "n3d_convT_bwd_weight")
    else:
        check(lib.n3d_conv_bwd_weight(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(dw), ptr(dbias), flags, ptr(in_gate),
                                      ptr(ws), n, jp, sp), "n3d_conv_bwd_weight")
    if job is not None and job.nchunks > 0:
        _ctx.final.append(job)
        _ctx.keep.append(ws)  # the partial slabs live in ws until StepContext.flush_final()


def conv_k1_norm_ok(g):
    return bool(_lib.load().n3d_conv_k1_norm_ok(C.byref(g)))


def conv_k1_norm_fwd(g, x: View, w, bias, y, oscale=None, oshift=None, stats=None):
    """y None: statistics of raw = W x + bias only (nothing stored); else y = oscale * raw + oshift in one pass over x"""
    flags = (_lib.SRC_BF16 if x.dt == _lib.BF16 else 0) | (_lib.DST_BF16 if (y is not None and y.dt == _lib.BF16) else 0)
    ws, wsp, n, flags = _packed(w, g, False, flags, x.t.device)
    check(_lib.load().n3d_conv_k1_norm_fwd(C.byref(g), x.p, x.ld, ptr(w), ptr(bias), y.p if y is not None else None, y.ld if y is not None else 0,
                                           flags, ptr(oscale), ptr(oshift), ptr(stats), wsp, n, stream_ptr()), "n3d_conv_k1_norm_fwd")


def conv_k1_norm_bwd_reduce(g, x: View, w, bias, dout: View, a, b, relu=False):
    """-> (sums, rows): the rows n3d_gn_bwd_coeffs reads, with raw recomputed from x"""
    lib = _lib.load()
    rows = int(lib.n3d_conv_k1_norm_rows(C.byref(g)))
    sums = torch.empty((x.B, rows, g.Co, 3), dtype=torch.float64, device=x.t.device)
    flags = _cflags(RELU if relu else 0, x, dout)
    check(lib.n3d_conv_k1_norm_bwd_reduce(C.byref(g), x.p, x.ld, ptr(w), ptr(bias), dout.p, dout.ld, ptr(a), ptr(b), flags, ptr(sums), stream_ptr()),
          "n3d_conv_k1_norm_bwd_reduce")
    return sums, rows


def conv_k1_norm_bwd_apply_wgrad(g, x: View, w, bias, dout: View, a, b, A, Bc, Cc_, dw, relu=False):
    """d(raw) in registers -> weight-gradient slabs (a weight-gradient launch like conv_bwd_weight: deferred reduction, queued for the
    side stream while the side-stream schedule is collecting)"""
    lib = _lib.load()
    flags = _cflags(RELU if relu else 0, x, dout)
    n = int(lib.n3d_conv_k1_norm_rows(C.byref(g))) * x.B * g.Ci * g.Co * 4     # one [Ci][Co] slab per workgroup
    ws = torch.empty(max(n, 256), dtype=torch.uint8, device=x.t.device)
    job = FinalJob() if _ctx is not None else None
    jp = C.byref(job) if job is not None else None

    def launch(sp):
        check(lib.n3d_conv_k1_norm_bwd_apply_wgrad(C.byref(g), x.p, x.ld, ptr(w), ptr(bias), dout.p, dout.ld, ptr(a), ptr(b), ptr(A), ptr(Bc), ptr(Cc_),
                                                   flags, ptr(dw), ptr(ws), n, jp, sp), "n3d_conv_k1_norm_bwd_apply_wgrad")

    if job is not None and _ctx.defer_wgrad:
        _ctx.wq.append((launch, job, ws, (x.t, dout.t, a, b, A, Bc, Cc_, dw, w, bias)))
        return
    launch(stream_ptr())
    if job is not None and job.nchunks > 0:
        _ctx.final.append(job)
        _ctx.keep.append(ws)


def conv_fwd2(calls):
    """Two forward convs, one launch where libn3d can fold them.  calls = [(g, x, w, bias, y, flags, in_gate, stats, transposed)] * 2"""
    cs, keep = [], []
    for (g, x, w, bias, y, flags, in_gate, stats, transposed) in calls:
        flags = _cflags(flags, x, y)
        ws, wsp, n, flags = _packed(w, g, transposed, flags, x.t.device)
        keep.append((ws, g))
        cs.append(ConvFwdCall(C.pointer(g), 1 if transposed else 0, flags, x.p.value, x.ld, w.data_ptr(), _vp(bias), y.p.value, y.ld,
                              _vp(in_gate), _vp(stats), wsp.value if hasattr(wsp, "value") else wsp, n))
    check(_lib.load().n3d_conv_fwd2(C.byref(cs[0]), C.byref(cs[1]), stream_ptr()), "n3d_conv_fwd2")


def conv_fwdN(calls):
    """Up to four forward convs, as few launches as libn3d can fold them into.  calls as conv_fwd2."""
    n = len(calls)
    arr = (ConvFwdCall * n)()
    keep = []
    for i, (g, x, w, bias, y, flags, in_gate, stats, transposed) in enumerate(calls):
        flags = _cflags(flags, x, y)
        ws, wsp, nb, flags = _packed(w, g, transposed, flags, x.t.device)
        keep.append((ws, g))
        arr[i] = ConvFwdCall(C.pointer(g), 1 if transposed else 0, flags, x.p.value, x.ld, w.data_ptr(), _vp(bias), y.p.value, y.ld,
                             _vp(in_gate), _vp(stats), wsp.value if hasattr(wsp, "value") else wsp, nb)
    check(_lib.load().n3d_conv_fwdN(arr, n, stream_ptr()), "n3d_conv_fwdN")


def conv_bwd_both2(calls):
    """Backward (data + weight gradient) of two convs, one launch where libn3d can fold them.
    calls = [(g, x, dy, w, dx, dw, dbias, flags_data, relu_src, out_gate, flags_weight, in_gate, transposed)] * 2"""
    if deferring():
        # side-stream schedule: only the data gradients stay on the backward chain (one launch where libn3d can fold them);
        # the two weight gradients are queued
        for (g, x, dy, w, dx, dw, dbias, flags_data, relu_src, out_gate, flags_weight, in_gate, transposed) in calls:
            conv_bwd_weight(g, x, dy, dw, dbias, flags_weight, in_gate, transposed)
        conv_bwd_data2([(g, dy, w, dx, flags_data, relu_src, out_gate, transposed)
                        for (g, x, dy, w, dx, dw, dbias, flags_data, relu_src, out_gate, flags_weight, in_gate, transposed) in calls])
        return
    cs, keep, jobs = [], [], []
    for (g, x, dy, w, dx, dw, dbias, flags_data, relu_src, out_gate, flags_weight, in_gate, transposed) in calls:
        _need_f32("conv_bwd_both2", x, dy, dx)
        flags_data, flags_weight = _mm(flags_data), _mm(flags_weight)
        wsd, wspd, nd, flags_data = _packed(w, g, not transposed, flags_data, dy.t.device)
        ws, n = _ws(g, x.t.device)
        job = FinalJob() if _ctx is not None else None
        keep.append((wsd, ws, g))
        jobs.append((job, ws))
        cs.append(ConvBwdCall(C.pointer(g), 1 if transposed else 0, flags_data, flags_weight, 0, x.p.value, x.ld, dy.p.value, dy.ld,
                              w.data_ptr(), dx.p.value, dx.ld, relu_src.p.value if relu_src is not None else None,
                              relu_src.ld if relu_src is not None else 0, _vp(out_gate),
                              wspd.value if hasattr(wspd, "value") else wspd, nd, _vp(dw), _vp(dbias), _vp(in_gate), ws.data_ptr(), n,
                              C.pointer(job) if job is not None else None))
    check(_lib.load().n3d_conv_bwd_both2(C.byref(cs[0]), C.byref(cs[1]), stream_ptr()), "n3d_conv_bwd_both2")
    for job, ws in jobs:
        if job is not None and job.nchunks > 0:
            _ctx.final.append(job)
            _ctx.keep.append(ws)


def dwconv_batch(jobs):
    """Depthwise gather passes of up to 8 primitives in one launch: jobs = [(g, data_grad, src View, w, bias | None, dst View, flags)];
    data_grad False: dst = conv(src) + bias, True: dst (+)= conv^T(src).  Same batch, channels and destination shape; distinct dsts."""
    n = len(jobs)
    arr = (DwJob * n)()
    for i, (g, data_grad, src, w, bias, dst, flags) in enumerate(jobs):
        _need_f32("dwconv_batch", src, dst)
        arr[i] = DwJob(C.pointer(g), 1 if data_grad else 0, flags, src.p.value, src.ld, w.data_ptr(), _vp(bias), dst.p.value, dst.ld)
    check(_lib.load().n3d_dwconv_batch(arr, n, stream_ptr()), "n3d_dwconv_batch")


def conv_bwd_data2(calls):
    """Data gradients of two convs, one launch where libn3d can fold them (distinct dx targets, small-tensor MFMA shapes).
    calls = [(g, dy, w, dx, flags, relu_src, out_gate, transposed)] * 2, arguments as conv_bwd_data."""
    cs, keep = [], []
    for (g, dy, w, dx, flags, relu_src, out_gate, transposed) in calls:
        _need_f32("conv_bwd_data2", dy, dx)
        ws, wsp, n, flags = _packed(w, g, not transposed, _mm(flags), dy.t.device)
        keep.append((ws, g))
        cs.append(ConvBwdCall(C.pointer(g), 1 if transposed else 0, flags, 0, 0, None, 0, dy.p.value, dy.ld, w.data_ptr(), dx.p.value, dx.ld,
                              relu_src.p.value if relu_src is not None else None, relu_src.ld if relu_src is not None else 0,
                              _vp(out_gate), wsp.value if hasattr(wsp, "value") else wsp, n, None, None, None, None, 0, None))
    check(_lib.load().n3d_conv_bwd_data2(C.byref(cs[0]), C.byref(cs[1]), stream_ptr()), "n3d_conv_bwd_data2")


def conv_bwd_both(g, x: View, dy: View, w, dx: View, dw, dbias, flags_data=0, relu_src: View | None = None, out_gate=None,
                  flags_weight=0, in_gate=None, transposed=False):
    """conv_bwd_data + conv_bwd_weight of one conv; one launch where libn3d can fold them."""
    _need_f32("conv_bwd_both", x, dy, dx)
    if deferring() and not g.depthwise:
        conv_bwd_weight(g, x, dy, dw, dbias, flags_weight, in_gate, transposed)      # queued for the side stream
        conv_bwd_data(g, dy, w, dx, flags_data, relu_src, out_gate, transposed)
        return
    flags_data, flags_weight = _mm(flags_data), _mm(flags_weight)
    wsd, wspd, nd, flags_data = _packed(w, g, not transposed, flags_data, dy.t.device)
    ws, n = _ws(g, x.t.device)
    job = FinalJob() if (_ctx is not None and not g.depthwise) else None
    jp = C.byref(job) if job is not None else None
    if transposed:
        if relu_src is not None or out_gate is not None or in_gate is not None or dbias is not None:
            raise N3DError("convT_bwd_both: relu / gate / bias-gradient extras are not supported")
        check(_lib.load().n3d_convT_bwd_both(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(w), dx.p, dx.ld, flags_data, wspd, nd,
                                             ptr(dw), flags_weight, ptr(ws), n, jp, stream_ptr()), "n3d_convT_bwd_both")
        if job is not None and job.nchunks > 0:
            _ctx.final.append(job)
            _ctx.keep.append(ws)
        return
    check(_lib.load().n3d_conv_bwd_both(C.byref(g), x.p, x.ld, dy.p, dy.ld, ptr(w), dx.p, dx.ld, flags_data,
                                        relu_src.p if relu_src is not None else None,
                                        relu_src.ld if relu_src is not None else 0, ptr(out_gate), wspd, nd,
                                        ptr(dw), ptr(dbias), flags_weight, ptr(in_gate), ptr(ws), n, jp, stream_ptr()),
          "n3d_conv_bwd_both")
    if job is not None and job.nchunks > 0:
        _ctx.final.append(job)
        _ctx.keep.append(ws)


# ------------------------------------------------------------------------------------------ epilogue
_stats_cache = None


class stats_cache:
    """with stats_cache(): channel statistics of one tensor are computed once -- inside a supernet cell the same input
    feeds several primitives that each need them (identity's GroupNorm, the SE gates).  Only valid while the tensors do
    not change: fused._run_forward holds it for one cell forward, whose inputs are complete before they are read."""

    def __enter__(self):
        global _stats_cache
        self.prev, _stats_cache = _stats_cache, {}
        return self

    def __exit__(self, *exc):
        global _stats_cache
        _stats_cache = self.prev
        return False


def channel_stats(x: View):
    if _stats_cache is not None:
        key = (x.p.value, x.ld, x.B, x.C, x.N, _redirect)    # per launch stream: a result computed on one stream is not ordered for the other
        hit = _stats_cache.get(key)
        if hit is None:
            hit = _stats_cache[key] = (_channel_stats(x), x.t)   # the tensor reference keeps its address from being reused
        return hit[0]
    return _channel_stats(x)


def channel_statsN(xs):
    """channel_stats of several tensors (cache honoured); the ones still missing that share a shape go out in one launch"""
    _need_f32("channel_statsN", *xs)
    out = [None] * len(xs)
    todo = {}
    for i, x in enumerate(xs):
        key = (x.p.value, x.ld, x.B, x.C, x.N, _redirect)
        hit = _stats_cache.get(key) if _stats_cache is not None else None
        if hit is not None:
            out[i] = hit[0]
        else:
            todo.setdefault((x.B, x.C, x.N), {}).setdefault(key, []).append(i)
    for (B, Cc, N), keys in todo.items():
        group = list(keys.items())
        for g0 in range(0, len(group), 8):
            chunk = group[g0:g0 + 8]
            views = [xs[idx[0]] for _, idx in chunk]
            if len(chunk) == 1:
                res = [_channel_stats(views[0])]
            else:
                rows = stats_rows(N, Cc)
                st = torch.empty((len(chunk), B, rows, Cc, 2), dtype=torch.float64, device=views[0].t.device)
                n = len(chunk)
                xp = (C.c_void_p * n)(*[v.p.value for v in views])
                lp = (C.c_int64 * n)(*[v.ld for v in views])
                sp = (C.c_void_p * n)(*[st[k].data_ptr() for k in range(n)])
                check(_lib.load().n3d_channel_statsN(xp, lp, sp, n, B, N, Cc, stream_ptr()), "n3d_channel_statsN")
                res = [(st[k], rows) for k in range(n)]
            for (key, idx), r, v in zip(chunk, res, views):
                if _stats_cache is not None:
                    _stats_cache[key] = (r, v.t)
                for i in idx:
                    out[i] = r
    return out


def _channel_stats(x: View):
    rows = stats_rows(x.N, x.C)
    st = torch.empty((x.B, rows, x.C, 2), dtype=torch.float64, device=x.t.device)
    check(_lib.load().n3d_channel_stats_t(x.p, x.ld, x.dt, x.B, x.N, x.C, ptr(st), stream_ptr()), "n3d_channel_stats_t")
    return st, rows


def _ng(G):
    """number of GroupNorm groups of a group-count argument: G < 0 is ONE group of -G real channels inside zero-padded ones (include/n3d.h,
    "padded channels"; the C layer takes the negative value as it is)"""
    return 1 if G < 0 else G


def gn_coeffs(stats, rows, gamma, beta, B, Cc, G, N, eps=1e-5):
    dev = stats.device
    a = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    b = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    mr = torch.empty((B, _ng(G), 2), dtype=torch.float32, device=dev)
    sr = torch.empty((B, Cc), dtype=torch.float64, device=dev)
    check(_lib.load().n3d_gn_coeffs(ptr(stats), rows, ptr(gamma), ptr(beta), B, Cc, G, N, eps, ptr(a), ptr(b), ptr(mr),
                                    ptr(sr), stream_ptr()), "n3d_gn_coeffs")
    return a, b, mr, sr


def affine_act(raw: View, a, b, wptr, out: View, flags=0):
    _same_dt("affine_act", raw, out)
    flags |= _aflag(raw)
    check(_lib.load().n3d_affine_act(raw.p, raw.ld, ptr(a), ptr(b), wptr, out.p, out.ld, raw.B, raw.N, raw.C, flags,
                                     stream_ptr()), "n3d_affine_act")


_FUSED_MAX_ROWS = None


def fused_max_rows():
    global _FUSED_MAX_ROWS
    if _FUSED_MAX_ROWS is None:
        _FUSED_MAX_ROWS = int(_lib.load().n3d_fused_max_rows())
    return _FUSED_MAX_ROWS


def affine_act_gn(raw: View, stats, rows, gamma, beta, G, eps, wptr, out: View, flags=0):
    """GroupNorm coefficients + normalise/activate/accumulate in one launch (small tensors)."""
    _same_dt("affine_act_gn", raw, out)
    flags |= _aflag(raw)
    dev = raw.t.device
    a = torch.empty((raw.B, raw.C), dtype=torch.float32, device=dev)
    b = torch.empty((raw.B, raw.C), dtype=torch.float32, device=dev)
    mr = torch.empty((raw.B, _ng(G), 2), dtype=torch.float32, device=dev)
    sr = torch.empty((raw.B, raw.C), dtype=torch.float64, device=dev)
    check(_lib.load().n3d_affine_act_gn(raw.p, raw.ld, ptr(stats), rows, ptr(gamma), ptr(beta), G, eps, wptr, out.p, out.ld,
                                        raw.B, raw.N, raw.C, flags, ptr(a), ptr(b), ptr(mr), ptr(sr), stream_ptr()),
          "n3d_affine_act_gn")
    return a, b, mr, sr


def affine_act_bwd_apply_gn(dout: View, raw: View, a, b, sums, rows, gamma, beta, mean_rstd, wptr, sumraw, conv_bias,
                            draw: View, G, flags=0, dalpha_ptr=None):
    dgamma = grad_target(gamma)
    dbeta = grad_target(beta)
    dcb = grad_target(conv_bias) if (conv_bias is not None and sumraw is not None) else None
    _same_dt("affine_act_bwd_apply_gn", dout, raw, draw)
    flags |= _aflag(raw)
    check(_lib.load().n3d_affine_act_bwd_apply_gn(dout.p, dout.ld, raw.p, raw.ld, ptr(a), ptr(b), ptr(sums), rows, ptr(gamma),
                                                  ptr(mean_rstd), wptr, ptr(sumraw), draw.p, draw.ld, raw.B, raw.N,
                                                  raw.C, G, flags, ptr(dgamma), ptr(dbeta), dalpha_ptr, ptr(dcb),
                                                  stream_ptr()), "n3d_affine_act_bwd_apply_gn")
    return dgamma, dbeta, dcb


def _vp(t):
    """raw device address (int or None) of a tensor / ctypes pointer, for structure fields"""
    if t is None:
        return None
    if isinstance(t, torch.Tensor):
        return t.data_ptr()
    return t.value if hasattr(t, "value") else t


def pair_ok(Cc, G, rows0, rows1, B):
    """shapes the FUSED node-level pair kernels accept (include/n3d.h, n3d_affine_act_gn2)"""
    G = _ng(G)
    return (4 <= Cc <= 64 and (Cc & (Cc - 1)) == 0 and Cc % G == 0 and Cc // G <= 16 and 1 <= rows0 <= fused_max_rows()
            and 1 <= rows1 <= fused_max_rows() and B <= 4)


def pair_shape_ok(Cc):
    """shapes every pair kernel accepts (the variants with separately computed coefficients have no row limit)"""
    return 4 <= Cc <= 64 and (Cc & (Cc - 1)) == 0


def affine_act_gn2(terms, G, eps, out: View, flags=0, out1: View | None = None):
    """Two GroupNorm -> [ReLU] -> weighted-sum epilogues into one output: terms = [(raw, stats, rows, gamma, beta, wptr, relu)] * 2.
    Returns [(a, b, mean_rstd, sumraw)] * 2 (saved for backward)."""
    raw0 = terms[0][0]
    _same_dt("affine_act_gn2", raw0, terms[1][0], out, out1)
    flags |= _aflag(raw0)
    dev = raw0.t.device
    B, Cc = raw0.B, raw0.C
    # one allocation for both terms' saved coefficients
    fbuf = torch.empty((2, 2 * B * Cc + 2 * B * _ng(G)), dtype=torch.float32, device=dev)
    dbuf = torch.empty((2, B * Cc), dtype=torch.float64, device=dev)
    ts, saved = [], []
    for i, (raw, stats, rows, gamma, beta, wptr, relu) in enumerate(terms):
        a = fbuf[i, :B * Cc].view(B, Cc)
        b = fbuf[i, B * Cc:2 * B * Cc].view(B, Cc)
        mr = fbuf[i, 2 * B * Cc:].view(B, _ng(G), 2)
        sr = dbuf[i].view(B, Cc)
        ts.append(GnFwdTerm(raw.p.value, raw.ld, stats.data_ptr(), rows, 1 if relu else 0, gamma.data_ptr(), beta.data_ptr(),
                            _vp(wptr), a.data_ptr(), b.data_ptr(), mr.data_ptr(), sr.data_ptr(), raw.dt, 0))
        saved.append((a, b, mr, sr))
    lib = _lib.load()
    o1p, o1ld = (out1.p, out1.ld) if out1 is not None else (None, 0)
    if pair_ok(Cc, G, terms[0][2], terms[1][2], B):
        check(lib.n3d_affine_act_gn2(C.byref(ts[0]), C.byref(ts[1]), G, eps, out.p, out.ld, o1p, o1ld, B, raw0.N, Cc, flags,
                                     stream_ptr()), "n3d_affine_act_gn2")
        return saved
    # large tensors: coefficients of both ops in one launch, then the two-term epilogue
    check(lib.n3d_gn_coeffs2(C.byref(ts[0]), C.byref(ts[1]), B, Cc, G, raw0.N, eps, stream_ptr()), "n3d_gn_coeffs2")
    check(lib.n3d_affine_act2(C.byref(ts[0]), C.byref(ts[1]), out.p, out.ld, o1p, o1ld, B, raw0.N, Cc, flags, stream_ptr()), "n3d_affine_act2")
    return saved


SMALL_NODE_BACKWARD = True   # the one-launch epilogue backward of a node on the small levels (n3d_affine_act_bwd_small2)


def affine_act_bwd_gn2(dout: View, terms, G, dout1: View | None = None):
    """Backward of affine_act_gn2: terms = [dict(raw, a, b, mr, sumraw, gamma, beta, wptr, relu, conv_bias, draw, dalpha_ptr)] * 2.
    Two launches (reduce, apply) for both ops.  Returns [(dgamma, dbeta, dconv_bias | None)] * 2."""
    raw0 = terms[0]["raw"]
    _same_dt("affine_act_bwd_gn2", raw0, terms[1]["raw"], terms[0]["draw"], terms[1]["draw"], dout, dout1)
    dev = raw0.t.device
    B, Cc, N = raw0.B, raw0.C, raw0.N
    lib = _lib.load()
    if SMALL_NODE_BACKWARD and all(t.get("dalpha_ptr") is None for t in terms) and small_backward_mode(B, N, Cc, G):
        # small levels: reduction, coefficients, parameter gradients and both d(raw) in ONE launch
        return affine_act_bwd_small(dout, terms, G, dout1)
    rows = stats_rows(N, Cc)
    sums = torch.empty((2, B, rows, Cc, 3), dtype=torch.float64, device=dev)
    ts, outs = [], []
    fused = pair_ok(Cc, G, rows, rows, B)
    coef = None if fused else torch.empty((2, 3, B, Cc), dtype=torch.float32, device=dev)
    for i, t in enumerate(terms):
        dgamma, dbeta = grad_target(t["gamma"]), grad_target(t["beta"])
        cb = t.get("conv_bias")
        dcb = grad_target(cb) if (cb is not None and t["sumraw"] is not None) else None
        raw, draw = t["raw"], t["draw"]
        ts.append(GnBwdTerm(raw.p.value, raw.ld, t["a"].data_ptr(), t["b"].data_ptr(), sums[i].data_ptr(), rows, 1 if t["relu"] else 0,
                            t["gamma"].data_ptr(), t["mr"].data_ptr(), _vp(t.get("wptr")), _vp(t["sumraw"]), draw.p.value, draw.ld,
                            _vp(dgamma), _vp(dbeta), _vp(t.get("dalpha_ptr")), _vp(dcb),
                            *([None] * 3 if fused else [coef[i, j].data_ptr() for j in range(3)]), raw.dt, 0))
        outs.append((dgamma, dbeta, dcb))
    lib = _lib.load()
    d1p, d1ld = (dout1.p, dout1.ld) if dout1 is not None else (None, 0)
    check(lib.n3d_affine_act_bwd_reduce2(dout.p, dout.ld, d1p, d1ld, C.byref(ts[0]), C.byref(ts[1]), B, N, Cc, stream_ptr()),
          "n3d_affine_act_bwd_reduce2")
    if fused:
        check(lib.n3d_affine_act_bwd_apply_gn2(dout.p, dout.ld, d1p, d1ld, C.byref(ts[0]), C.byref(ts[1]), B, N, Cc, G, stream_ptr()),
              "n3d_affine_act_bwd_apply_gn2")
    else:
        check(lib.n3d_gn_bwd_coeffs2(C.byref(ts[0]), C.byref(ts[1]), B, Cc, G, N, stream_ptr()), "n3d_gn_bwd_coeffs2")
        check(lib.n3d_affine_act_bwd_apply2(dout.p, dout.ld, d1p, d1ld, C.byref(ts[0]), C.byref(ts[1]), B, N, Cc, stream_ptr()),
              "n3d_affine_act_bwd_apply2")
    return outs


_ticket_pools = {}   # device index -> [zeroed int32 words, next word]


def _tickets(device, n):
    """`n` zeroed words for a kernel that draws self-resetting tickets (atomicInc wrapping at the last one: zero again when the
    launch is over, so a replayed graph and the next call site that comes round to the same words find them ready).  Two launches
    may not use the same words AT THE SAME TIME; the pool hands out 4096 words round-robin, a step uses a few dozen."""
    key = device.index or 0
    pool = _ticket_pools.get(key)
    if pool is None:
        pool = _ticket_pools[key] = [torch.zeros(4096, dtype=torch.int32, device=device), 0]
    if pool[1] + n > 4096:
        pool[1] = 0
    ptr = pool[0].data_ptr() + 4 * pool[1]
    pool[1] += n
    return ptr


_small_modes = {}
# (mode 2 of n3d_bwd_small_mode -- the 8^3 level at batch 2, one workgroup per (group, sample) -- is not used: on the benchmarked step it
# measured no faster than the reduce2 + apply_gn2 pair it replaces, main chain 1.928 vs 1.920 ms, profiles/r03_contention_probes.log)


def small_backward_mode(B, N, Cc, G):
    """0 = the one-launch epilogue backward does not take this shape; 1 = one workgroup per GroupNorm group (include/n3d.h,
    n3d_affine_act_bwd_small)"""
    key = (B, N, Cc, G)
    m = _small_modes.get(key)
    if m is None:
        m = _small_modes[key] = int(_lib.load().n3d_bwd_small2_ok(B, N, Cc, G))
    return m


def affine_act_bwd_small(dout: View, terms, G, dout1: View | None = None):
    """The whole GroupNorm-epilogue backward of one or two terms (dicts as affine_act_bwd_gn2) in ONE launch; the caller has checked
    small_backward_mode().  Returns [(dgamma, dbeta, dconv_bias | None)] per term."""
    raw0 = terms[0]["raw"]
    dev = raw0.t.device
    B, Cc, N = raw0.B, raw0.C, raw0.N
    lib = _lib.load()
    ts, outs = [], []
    for t in terms:
        dgamma, dbeta = grad_target(t["gamma"]), grad_target(t["beta"])
        cb = t.get("conv_bias")
        dcb = grad_target(cb) if (cb is not None and t["sumraw"] is not None) else None
        raw, draw = t["raw"], t["draw"]
        ts.append(GnBwdTerm(raw.p.value, raw.ld, t["a"].data_ptr(), t["b"].data_ptr(), None, 0, 1 if t["relu"] else 0,
                            t["gamma"].data_ptr(), t["mr"].data_ptr(), _vp(t.get("wptr")), _vp(t["sumraw"]), draw.p.value, draw.ld,
                            _vp(dgamma), _vp(dbeta), None, _vp(dcb), None, None, None, raw.dt, 0))
        outs.append((dgamma, dbeta, dcb))
    d1p, d1ld = (dout1.p, dout1.ld) if dout1 is not None else (None, 0)
    scratch, sbytes, tick = None, 0, None
    if small_backward_mode(B, N, Cc, G) == 2:
        sbytes = int(lib.n3d_bwd_small_scratch_bytes(B, G))
        scratch = torch.empty(sbytes // 8, dtype=torch.float64, device=dev)
        tick = _tickets(dev, G)
    check(lib.n3d_affine_act_bwd_small(dout.p, dout.ld, d1p, d1ld, C.byref(ts[0]), C.byref(ts[1]) if len(ts) > 1 else None, B, N, Cc, G,
                                       scratch.data_ptr() if scratch is not None else None, sbytes, tick, stream_ptr()),
          "n3d_affine_act_bwd_small")
    return outs


MAX_GROUP_TERMS = 8  # N3D_MAX_GROUP_TERMS


def group_shape_ok(Cc):
    """channel counts the N-term epilogues accept (include/n3d.h, n3d_affine_actN)"""
    return 4 <= Cc <= 64 and (Cc & (Cc - 1)) == 0


def gn_coeffsN(terms, G, eps):
    """GroupNorm coefficients of up to 8 tensors of one shape in ONE launch: terms = [(raw, stats, rows, gamma, beta)].
    Returns [(a, b, mean_rstd, sumraw)] (what affine_actN reads and backward needs)."""
    n = len(terms)
    raw0 = terms[0][0]
    _need_f32("gn_coeffsN / affine_actN", *[t[0] for t in terms])
    dev = raw0.t.device
    B, Cc = raw0.B, raw0.C
    fbuf = torch.empty((n, 2 * B * Cc + 2 * B * _ng(G)), dtype=torch.float32, device=dev)
    dbuf = torch.empty((n, B * Cc), dtype=torch.float64, device=dev)
    arr = (GnFwdTerm * n)()
    saved = []
    for i, (raw, stats, rows, gamma, beta) in enumerate(terms):
        a = fbuf[i, :B * Cc].view(B, Cc)
        b = fbuf[i, B * Cc:2 * B * Cc].view(B, Cc)
        mr = fbuf[i, 2 * B * Cc:].view(B, _ng(G), 2)
        sr = dbuf[i].view(B, Cc)
        arr[i] = GnFwdTerm(raw.p.value, raw.ld, stats.data_ptr(), rows, 0, gamma.data_ptr(), beta.data_ptr(), None, a.data_ptr(),
                           b.data_ptr(), mr.data_ptr(), sr.data_ptr())
        saved.append((a, b, mr, sr))
    check(_lib.load().n3d_gn_coeffsN(arr, n, B, Cc, G, raw0.N, eps, stream_ptr()), "n3d_gn_coeffsN")
    return saved


def node_fwd_coeffs(gn_terms, G, eps, se_terms):
    """gn_coeffsN(gn_terms, G, eps) and se_gate_fwdN(se_terms) -- the coefficient computations of one supernet node group -- in ONE
    launch: gn_terms = [(raw, stats, rows, gamma, beta)], se_terms = [(stats, rows, fc)] on tensors of the same (B, C, N).
    Returns ([(a, b, mean_rstd, sumraw)], [(mean, hidden, gate)])."""
    n, m = len(gn_terms), len(se_terms)
    raw0 = gn_terms[0][0]
    _need_f32("node_fwd_coeffs", *[t[0] for t in gn_terms])
    dev = raw0.t.device
    B, Cc = raw0.B, raw0.C
    fbuf = torch.empty((n, 2 * B * Cc + 2 * B * _ng(G)), dtype=torch.float32, device=dev)
    dbuf = torch.empty((n, B * Cc), dtype=torch.float64, device=dev)
    arr = (GnFwdTerm * n)()
    saved = []
    for i, (raw, stats, rows, gamma, beta) in enumerate(gn_terms):
        a = fbuf[i, :B * Cc].view(B, Cc)
        b = fbuf[i, B * Cc:2 * B * Cc].view(B, Cc)
        mr = fbuf[i, 2 * B * Cc:].view(B, _ng(G), 2)
        sr = dbuf[i].view(B, Cc)
        arr[i] = GnFwdTerm(raw.p.value, raw.ld, stats.data_ptr(), rows, 0, gamma.data_ptr(), beta.data_ptr(), None, a.data_ptr(),
                           b.data_ptr(), mr.data_ptr(), sr.data_ptr())
        saved.append((a, b, mr, sr))
    buf = torch.empty((m, 2 * B * Cc + B), dtype=torch.float32, device=dev)
    sarr = (SeTerm * m)()
    out = []
    for i, (stats, rows, fc) in enumerate(se_terms):
        mean, gate, hidden = buf[i, :B * Cc].view(B, Cc), buf[i, B * Cc:2 * B * Cc].view(B, Cc), buf[i, 2 * B * Cc:]
        sarr[i] = SeTerm(stats.data_ptr(), rows, 0, fc[0].weight.data_ptr(), fc[0].bias.data_ptr(), fc[2].weight.data_ptr(),
                         fc[2].bias.data_ptr(), mean.data_ptr(), hidden.data_ptr(), gate.data_ptr(), *([None] * 8))
        out.append((mean, hidden, gate))
    check(_lib.load().n3d_node_fwd_coeffs(arr, n, sarr, m, B, Cc, G, raw0.N, eps, stream_ptr()), "n3d_node_fwd_coeffs")
    return saved, out


def affine_actN(terms, out: View, flags=0):
    """out (+)= sum_k w_k * act_k(a_k * raw_k + b_k) for up to 8 terms in one pass: terms = [(raw, a | None, b | None, wptr, relu)]."""
    n = len(terms)
    raw0 = terms[0][0]
    _need_f32("affine_actN", out, *[t[0] for t in terms])
    arr = (GnFwdTerm * n)()
    for i, (raw, a, b, wptr, relu) in enumerate(terms):
        arr[i] = GnFwdTerm(raw.p.value, raw.ld, None, 0, 1 if relu else 0, None, None, _vp(wptr), _vp(a), _vp(b), None, None)
    check(_lib.load().n3d_affine_actN(arr, n, out.p, out.ld, raw0.B, raw0.N, raw0.C, flags, stream_ptr()), "n3d_affine_actN")


class GnGroupBwd:
    """Backward of one N-term GroupNorm group (affine_act_gnN), phase by phase: reductions, coefficients + parameter gradients, d(raw).
    terms = [dict(raw, a, b, mr, sumraw, gamma, beta, wptr, relu, conv_bias, draw, dalpha_ptr)]; .outs = [(dgamma, dbeta, dconv_bias |
    None)].  The phases of several groups (and of a node's other primitives) can share launches: node_bwd_prologue."""

    def __init__(self, dout: View, terms, G):
        n = len(terms)
        raw0 = terms[0]["raw"]
        _need_f32("affine_act_bwd_gnN", dout, *[t["raw"] for t in terms])
        dev = raw0.t.device
        self.dout, self.terms, self.G, self.n = dout, terms, G, n
        self.B, self.C, self.N = raw0.B, raw0.C, raw0.N
        rows = stats_rows(self.N, self.C)
        self.sums = torch.empty((n, self.B, rows, self.C, 3), dtype=torch.float64, device=dev)
        self.coef = torch.empty((n, 3, self.B, self.C), dtype=torch.float32, device=dev)
        self.arr = (GnBwdTerm * n)()
        self.outs = []
        for i, t in enumerate(terms):
            dgamma, dbeta = grad_target(t["gamma"]), grad_target(t["beta"])
            cb = t.get("conv_bias")
            dcb = grad_target(cb) if (cb is not None and t["sumraw"] is not None) else None
            raw, draw = t["raw"], t["draw"]
            self.arr[i] = GnBwdTerm(raw.p.value, raw.ld, t["a"].data_ptr(), t["b"].data_ptr(), self.sums[i].data_ptr(), rows, 1 if t["relu"] else 0,
                                    t["gamma"].data_ptr(), t["mr"].data_ptr(), _vp(t.get("wptr")), _vp(t["sumraw"]), draw.p.value, draw.ld,
                                    _vp(dgamma), _vp(dbeta), _vp(t.get("dalpha_ptr")), _vp(dcb), *[self.coef[i, j].data_ptr() for j in range(3)])
            self.outs.append((dgamma, dbeta, dcb))

    def reduce(self):
        check(_lib.load().n3d_affine_act_bwd_reduceN(self.dout.p, self.dout.ld, self.arr, self.n, self.B, self.N, self.C, stream_ptr()),
              "n3d_affine_act_bwd_reduceN")

    def coeffs(self):
        check(_lib.load().n3d_gn_bwd_coeffsN(self.arr, self.n, self.B, self.C, self.G, self.N, stream_ptr()), "n3d_gn_bwd_coeffsN")

    def apply(self):
        if getattr(self, "applied", False):      # (ran in node_bwd_prologue's launch for all groups of the level)
            return
        check(_lib.load().n3d_affine_act_bwd_applyN(self.dout.p, self.dout.ld, self.arr, self.n, self.B, self.N, self.C, stream_ptr()),
              "n3d_affine_act_bwd_applyN")


def affine_act_bwd_gnN(dout: View, terms, G):
    """Backward of affine_act_gnN, three launches for all terms (reductions, coefficients + parameter gradients, d(raw)).
    Returns [(dgamma, dbeta, dconv_bias | None)]."""
    g = GnGroupBwd(dout, terms, G)
    g.reduce()
    g.coeffs()
    g.apply()
    return g.outs


MAX_REDUCE_TERMS = 16      # N3D_MAX_REDUCE_TERMS


def node_bwd_prologue(dout: View, groups, singles, gates, idents=()):
    """The reduction and coefficient phases of ONE node level of the supernet backward, two launches for everything that consumes the
    node gradient `dout`: groups = [GnGroupBwd] (their apply() is left to the caller), singles = [(raw, a | None, b | None, relu)] =
    the reductions of the node's other primitives (affine_act_bwd_reduceN), gates = [(index into singles, dict(wptr, mean, hidden,
    gate, fc, dalpha_ptr))] = the SE gates among them (se_gate_bwdN), idents = [(index into singles, dict(gamma, beta, mr, wptr,
    dalpha_ptr))] = identity-with-norm primitives whose GroupNorm coefficients join the groups' launch (they are dropped, i.e. left
    to their own fused kernel, when the launch has no room).  Returns ([(sums, rows)] per single, [(dw1, db1, dw2, db2, A, Bc)] per
    gate, [(dgamma, dbeta, cA, cB, cC) | None] per ident).  More than 16 reductions / 16 GroupNorm terms / 8 gates: the phases fall
    back to one launch per group."""
    lib = _lib.load()
    g0 = groups[0]
    B, Cc, N, dev = g0.B, g0.C, g0.N, dout.t.device
    rows = stats_rows(N, Cc)
    ns = len(singles)
    _need_f32("node_bwd_prologue", dout, *[t[0] for t in singles])
    ssum = torch.empty((max(ns, 1), B, rows, Cc, 3), dtype=torch.float64, device=dev)
    ngn = sum(g.n for g in groups)
    # ---- reductions: every term of the level, 16 per launch
    allt = [g.arr[i] for g in groups for i in range(g.n)]
    for i, (raw, a, b, relu) in enumerate(singles):
        allt.append(GnBwdTerm(raw.p.value, raw.ld, _vp(a), _vp(b), ssum[i].data_ptr(), rows, 1 if relu else 0, *([None] * 4), None, 0, *([None] * 7)))
    for i0 in range(0, len(allt), MAX_REDUCE_TERMS):
        chunk = allt[i0:i0 + MAX_REDUCE_TERMS]
        arr = (GnBwdTerm * len(chunk))(*chunk)
        check(lib.n3d_affine_act_bwd_reduceN(dout.p, dout.ld, arr, len(chunk), B, N, Cc, stream_ptr()), "n3d_affine_act_bwd_reduceN")
    pre = [(ssum[i], rows) for i in range(ns)]
    # ---- coefficients
    se_arr, se_out = None, []
    if gates:
        coef = torch.empty((len(gates), 2, B, Cc), dtype=torch.float32, device=dev)
        se_arr = (SeTerm * len(gates))()
        for i, (si, t) in enumerate(gates):
            fc = t["fc"]
            dw1, db1, dw2, db2 = (grad_target(fc[0].weight), grad_target(fc[0].bias), grad_target(fc[2].weight), grad_target(fc[2].bias))
            dw1 = dw1 if dw1 is not None else torch.empty((1, Cc), dtype=torch.float32, device=dev)
            db1 = db1 if db1 is not None else torch.empty((1,), dtype=torch.float32, device=dev)
            dw2 = dw2 if dw2 is not None else torch.empty((Cc, 1), dtype=torch.float32, device=dev)
            db2 = db2 if db2 is not None else torch.empty((Cc,), dtype=torch.float32, device=dev)
            se_arr[i] = SeTerm(ssum[si].data_ptr(), rows, 0, fc[0].weight.data_ptr(), None, fc[2].weight.data_ptr(), None,
                               t["mean"].data_ptr(), t["hidden"].data_ptr(), t["gate"].data_ptr(), _vp(t.get("wptr")), dw1.data_ptr(),
                               db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), _vp(t.get("dalpha_ptr")), coef[i, 0].data_ptr(), coef[i, 1].data_ptr())
            se_out.append((dw1, db1, dw2, db2, coef[i, 0], coef[i, 1]))
    same_g = len({g.G for g in groups}) == 1
    id_out = [None] * len(idents)
    if same_g and ngn <= MAX_REDUCE_TERMS and len(gates) <= MAX_GROUP_TERMS:
        nid = min(len(idents), MAX_REDUCE_TERMS - ngn)
        garr = (GnBwdTerm * (ngn + nid))()
        k = 0
        for g in groups:
            for i in range(g.n):
                garr[k] = g.arr[i]
                k += 1
        if nid:
            icoef = torch.empty((nid, 3, B, Cc), dtype=torch.float32, device=dev)
            for i, (si, t) in enumerate(idents[:nid]):
                dgamma, dbeta = grad_target(t["gamma"]), grad_target(t["beta"])
                dgamma = dgamma if dgamma is not None else torch.empty((Cc,), dtype=torch.float32, device=dev)
                dbeta = dbeta if dbeta is not None else torch.empty((Cc,), dtype=torch.float32, device=dev)
                garr[k + i] = GnBwdTerm(None, 0, None, None, ssum[si].data_ptr(), rows, 0, t["gamma"].data_ptr(), t["mr"].data_ptr(), _vp(t.get("wptr")),
                                        None, None, 0, dgamma.data_ptr(), dbeta.data_ptr(), _vp(t.get("dalpha_ptr")), None,
                                        *[icoef[i, j].data_ptr() for j in range(3)])
                id_out[i] = (dgamma, dbeta, icoef[i, 0], icoef[i, 1], icoef[i, 2])
        check(lib.n3d_node_bwd_coeffs(garr, ngn + nid, se_arr, len(gates), B, Cc, g0.G, N, stream_ptr()), "n3d_node_bwd_coeffs")
    else:
        for g in groups:
            g.coeffs()
        for i0 in range(0, len(gates), MAX_GROUP_TERMS):
            n = min(MAX_GROUP_TERMS, len(gates) - i0)
            sub = (SeTerm * n)(*[se_arr[i0 + i] for i in range(n)])
            check(lib.n3d_se_gate_bwdN(sub, n, N, B, Cc, stream_ptr()), "n3d_se_gate_bwdN")
    # ---- the groups' apply passes in ONE launch where they fit (their d(raw) buffers are what the weight ops read later)
    if len(groups) >= 2 and ngn <= MAX_REDUCE_TERMS:
        garr = (GnBwdTerm * ngn)(*[g.arr[i] for g in groups for i in range(g.n)])
        check(lib.n3d_affine_act_bwd_applyN(dout.p, dout.ld, garr, ngn, B, N, Cc, stream_ptr()), "n3d_affine_act_bwd_applyN")
        for g in groups:
            g.applied = True
    return pre, se_out, id_out


def node_bwd_apply_sum(dout: View, items):
    """The apply passes of a node level's single primitives in ONE launch: items = [(raw, a | None, b | None, relu, cA, cB, cC | None,
    target View, accumulate)] in issue order; consecutive items with the same target are summed into it in that order (the first one's
    `accumulate` decides whether on top of its previous content) -- what K.affine_act_bwd_apply calls in that order would leave there."""
    n = len(items)
    raw0 = items[0][0]
    _need_f32("node_bwd_apply_sum", dout, *[it[0] for it in items], *[it[7] for it in items])
    arr = (GnBwdTerm * n)()
    for i, (raw, a, b, relu, cA, cB, cC, target, acc) in enumerate(items):
        arr[i] = GnBwdTerm(raw.p.value, raw.ld, _vp(a), _vp(b), None, 0, 1 if relu else 0, None, None, None, None, target.p.value, target.ld,
                           None, None, None, None, _vp(cA), _vp(cB), _vp(cC), 0, 1 if acc else 0)
    check(_lib.load().n3d_affine_act_bwd_apply_sum(dout.p, dout.ld, arr, n, raw0.B, raw0.N, raw0.C, stream_ptr()), "n3d_affine_act_bwd_apply_sum")


def affine_act_bwd_reduceN(dout: View, terms):
    """affine_act_bwd_reduce of up to 8 terms that consume the same gradient, one launch: terms = [(raw, a | None, b | None, relu)].
    Returns [(sums, rows)]."""
    n = len(terms)
    raw0 = terms[0][0]
    _need_f32("affine_act_bwd_reduceN", dout, *[t[0] for t in terms])
    rows = stats_rows(raw0.N, raw0.C)
    sums = torch.empty((n, raw0.B, rows, raw0.C, 3), dtype=torch.float64, device=raw0.t.device)
    arr = (GnBwdTerm * n)()
    for i, (raw, a, b, relu) in enumerate(terms):
        arr[i] = GnBwdTerm(raw.p.value, raw.ld, _vp(a), _vp(b), sums[i].data_ptr(), rows, 1 if relu else 0, *([None] * 4), None, 0, *([None] * 7))
    check(_lib.load().n3d_affine_act_bwd_reduceN(dout.p, dout.ld, arr, n, raw0.B, raw0.N, raw0.C, stream_ptr()), "n3d_affine_act_bwd_reduceN")
    return [(sums[i], rows) for i in range(n)]


def affine_act_bwd_reduce(dout: View, raw: View, a, b, flags=0):
    _same_dt("affine_act_bwd_reduce", dout, raw)
    flags |= _aflag(raw)
    rows = stats_rows(raw.N, raw.C)
    sums = torch.empty((raw.B, rows, raw.C, 3), dtype=torch.float64, device=raw.t.device)
    check(_lib.load().n3d_affine_act_bwd_reduce(dout.p, dout.ld, raw.p, raw.ld, ptr(a), ptr(b), raw.B, raw.N, raw.C,
                                                flags, ptr(sums), stream_ptr()), "n3d_affine_act_bwd_reduce")
    return sums, rows


def grad_target(p):
    """Where the gradient of parameter `p` is written: its slice of a flat gradient buffer when a
    trainer installed one (`p._n3d_grad`, written in place, no autograd accumulation kernels), else a
    fresh tensor that the autograd Function returns."""
    if p is None or not p.requires_grad:
        return None
    t = getattr(p, "_n3d_grad", None)
    return t if t is not None else torch.empty_like(p)


def gn_bwd_coeffs(sums, rows, gamma, mean_rstd, wptr, B, Cc, G, N, dalpha_ptr=None, beta=None, sumraw=None, conv_bias=None):
    dev = sums.device
    dgamma = grad_target(gamma) if isinstance(gamma, torch.nn.Parameter) else torch.empty((Cc,), dtype=torch.float32, device=dev)
    dbeta = grad_target(beta) if isinstance(beta, torch.nn.Parameter) else torch.empty((Cc,), dtype=torch.float32, device=dev)
    A = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    Bc = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    Cc_ = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    dcb = grad_target(conv_bias) if (conv_bias is not None and sumraw is not None) else None
    check(_lib.load().n3d_gn_bwd_coeffs(ptr(sums), rows, ptr(gamma), ptr(mean_rstd), wptr, B, Cc, G, N, ptr(dgamma),
                                        ptr(dbeta), dalpha_ptr, ptr(A), ptr(Bc), ptr(Cc_), ptr(sumraw), ptr(dcb),
                                        stream_ptr()), "n3d_gn_bwd_coeffs")
    return dgamma, dbeta, A, Bc, Cc_, dcb


def plain_bwd_coeffs(sums, rows, wptr, B, Cc, device, dalpha_ptr=None, want_A=True):
    A = torch.empty((B, Cc), dtype=torch.float32, device=device) if want_A else None
    check(_lib.load().n3d_plain_bwd_coeffs(ptr(sums), rows, wptr, B, Cc, dalpha_ptr, ptr(A), stream_ptr()),
          "n3d_plain_bwd_coeffs")
    return A


def affine_act_bwd_apply(dout: View, raw: View, a, b, A, Bc, Cc_, draw: View, flags=0):
    _same_dt("affine_act_bwd_apply", dout, raw, draw)
    flags |= _aflag(raw)
    check(_lib.load().n3d_affine_act_bwd_apply(dout.p, dout.ld, raw.p, raw.ld, ptr(a), ptr(b), ptr(A), ptr(Bc), ptr(Cc_),
                                               draw.p, draw.ld, raw.B, raw.N, raw.C, flags, stream_ptr()),
          "n3d_affine_act_bwd_apply")


# ------------------------------------------------------------------------------------------ SE / pool
def se_gate_fwd(stats, rows, N, w1, b1, w2, b2, B, Cc):
    dev = stats.device
    mean = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    hidden = torch.empty((B,), dtype=torch.float32, device=dev)
    gate = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    check(_lib.load().n3d_se_gate_fwd(ptr(stats), rows, N, ptr(w1), ptr(b1), ptr(w2), ptr(b2), B, Cc, ptr(mean),
                                      ptr(hidden), ptr(gate), stream_ptr()), "n3d_se_gate_fwd")
    return mean, hidden, gate


def se_gate_bwd(sums, rows, wptr, mean, hidden, gate, w1, w2, B, Cc, N, dalpha_ptr=None, fc=None):
    dev = sums.device
    if fc is not None:
        dw1, db1, dw2, db2 = (grad_target(fc[0].weight), grad_target(fc[0].bias), grad_target(fc[2].weight),
                              grad_target(fc[2].bias))
    else:
        dw1 = db1 = dw2 = db2 = None
    dw1 = dw1 if dw1 is not None else torch.empty((1, Cc), dtype=torch.float32, device=dev)
    db1 = db1 if db1 is not None else torch.empty((1,), dtype=torch.float32, device=dev)
    dw2 = dw2 if dw2 is not None else torch.empty((Cc, 1), dtype=torch.float32, device=dev)
    db2 = db2 if db2 is not None else torch.empty((Cc,), dtype=torch.float32, device=dev)
    A = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    Bc = torch.empty((B, Cc), dtype=torch.float32, device=dev)
    check(_lib.load().n3d_se_gate_bwd(ptr(sums), rows, wptr, ptr(mean), ptr(hidden), ptr(gate), ptr(w1), ptr(w2), B, Cc,
                                      N, ptr(dw1), ptr(db1), ptr(dw2), ptr(db2), dalpha_ptr, ptr(A), ptr(Bc),
                                      stream_ptr()), "n3d_se_gate_bwd")
    return dw1, db1, dw2, db2, A, Bc


def se_gate_fwdN(terms, N, B, Cc):
    """se_gate_fwd of up to 8 gates in one launch: terms = [(stats, rows, fc)].  Returns [(mean, hidden, gate)]."""
    n = len(terms)
    dev = terms[0][0].device
    buf = torch.empty((n, 2 * B * Cc + B), dtype=torch.float32, device=dev)
    arr = (SeTerm * n)()
    out = []
    for i, (stats, rows, fc) in enumerate(terms):
        mean, gate, hidden = buf[i, :B * Cc].view(B, Cc), buf[i, B * Cc:2 * B * Cc].view(B, Cc), buf[i, 2 * B * Cc:]
        arr[i] = SeTerm(stats.data_ptr(), rows, 0, fc[0].weight.data_ptr(), fc[0].bias.data_ptr(), fc[2].weight.data_ptr(),
                        fc[2].bias.data_ptr(), mean.data_ptr(), hidden.data_ptr(), gate.data_ptr(), *([None] * 8))
        out.append((mean, hidden, gate))
    check(_lib.load().n3d_se_gate_fwdN(arr, n, N, B, Cc, stream_ptr()), "n3d_se_gate_fwdN")
    return out


def se_gate_bwdN(terms, N, B, Cc):
    """se_gate_bwd of up to 8 gates in one launch: terms = [dict(sums, rows, wptr, mean, hidden, gate, fc, dalpha_ptr)].
    Returns [(dw1, db1, dw2, db2, A, Bc)]."""
    n = len(terms)
    dev = terms[0]["sums"].device
    coef = torch.empty((n, 2, B, Cc), dtype=torch.float32, device=dev)
    arr = (SeTerm * n)()
    out = []
    for i, t in enumerate(terms):
        fc = t["fc"]
        dw1, db1, dw2, db2 = (grad_target(fc[0].weight), grad_target(fc[0].bias), grad_target(fc[2].weight), grad_target(fc[2].bias))
        dw1 = dw1 if dw1 is not None else torch.empty((1, Cc), dtype=torch.float32, device=dev)
        db1 = db1 if db1 is not None else torch.empty((1,), dtype=torch.float32, device=dev)
        dw2 = dw2 if dw2 is not None else torch.empty((Cc, 1), dtype=torch.float32, device=dev)
        db2 = db2 if db2 is not None else torch.empty((Cc,), dtype=torch.float32, device=dev)
        arr[i] = SeTerm(t["sums"].data_ptr(), t["rows"], 0, fc[0].weight.data_ptr(), None, fc[2].weight.data_ptr(), None,
                        t["mean"].data_ptr(), t["hidden"].data_ptr(), t["gate"].data_ptr(), _vp(t.get("wptr")), dw1.data_ptr(),
                        db1.data_ptr(), dw2.data_ptr(), db2.data_ptr(), _vp(t.get("dalpha_ptr")), coef[i, 0].data_ptr(), coef[i, 1].data_ptr())
        out.append((dw1, db1, dw2, db2, coef[i, 0], coef[i, 1]))
    check(_lib.load().n3d_se_gate_bwdN(arr, n, N, B, Cc, stream_ptr()), "n3d_se_gate_bwdN")
    return out


def pool2_fwd(x: View, y: View, is_max):
    _same_dt("pool2_fwd", x, y)
    check(_lib.load().n3d_pool2_fwd(x.p, x.ld, y.p, y.ld, x.B, x.D, x.H, x.W, x.C, (POOL_MAX if is_max else 0) | _aflag(x),
                                    stream_ptr()), "n3d_pool2_fwd")


def pool2_bwd(dy: View, x: View, dx: View, is_max, accumulate=False, wptr=None):
    """dx (+)= w * pool^T(dy); wptr: device scalar (the MixedOp weight of the pooling primitive) or None = 1"""
    fl = (POOL_MAX if is_max else 0) | (ACCUMULATE if accumulate else 0) | _aflag(dy)
    _same_dt("pool2_bwd", dy, x, dx)
    check(_lib.load().n3d_pool2_bwd_scaled(dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.B, x.D, x.H, x.W, x.C, fl, wptr, stream_ptr()),
          "n3d_pool2_bwd_scaled")


def pool2_fwd_both(x: View, y_avg: View, y_max: View):
    _need_f32("pool2_fwd_both", x, y_avg, y_max)
    check(_lib.load().n3d_pool2_fwd_both(x.p, x.ld, y_avg.p, y_avg.ld, y_max.p, y_max.ld, x.B, x.D, x.H, x.W, x.C, stream_ptr()),
          "n3d_pool2_fwd_both")


def pool2_bwd_both(dy: View, x: View, dx: View, accumulate, w_avg=None, w_max=None):
    """dx (+)= w_avg * avgpool^T(dy) + w_max * maxpool^T(dy); w_*: device scalar pointers or None = 1"""
    _need_f32("pool2_bwd_both", dy, x, dx)
    check(_lib.load().n3d_pool2_bwd_both(dy.p, dy.ld, x.p, x.ld, dx.p, dx.ld, x.B, x.D, x.H, x.W, x.C, ACCUMULATE if accumulate else 0,
                                         w_avg, w_max, stream_ptr()), "n3d_pool2_bwd_both")


def plain_dalphaN(terms, B, Cc):
    """dalpha = <dout, z> of up to 8 un-normalised primitives from their reduction rows, one launch:
    terms = [(sums, rows, dalpha_ptr)]"""
    n = len(terms)
    arr = (PlainCoefTerm * n)()
    for i, (sums, rows, dap) in enumerate(terms):
        arr[i] = PlainCoefTerm(sums.data_ptr(), rows, 0, None, _vp(dap), None)
    check(_lib.load().n3d_plain_bwd_coeffsN(arr, n, B, Cc, stream_ptr()), "n3d_plain_bwd_coeffsN")


# ------------------------------------------------------------------------------------------ dice / adam
def _bcv_strides(t):
    """(sb, sc, sv) element strides of a (B,C,D,H,W) tensor whose voxels are uniformly strided, or None."""
    B, Cc, D, H, W = t.shape
    s = t.stride()
    sv = s[4] if W > 1 else (s[3] if H > 1 else (s[2] if D > 1 else 1))
    if (H > 1 and s[3] != W * sv) or (D > 1 and s[2] != H * W * sv):
        return None
    return s[0], s[1], sv


def dice_fwd(p, t, smooth):
    lib = _lib.load()
    B, Cc = p.shape[0], p.shape[1]
    N = p.shape[2] * p.shape[3] * p.shape[4]
    ps, ts = _bcv_strides(p), _bcv_strides(t)
    rows = int(lib.n3d_dice_rows(N))
    partial = torch.empty((B, Cc, rows, 3), dtype=torch.float64, device=p.device)
    sums = torch.empty((B, Cc, 3), dtype=torch.float64, device=p.device)
    loss = torch.empty((), dtype=torch.float32, device=p.device)
    check(lib.n3d_dice_fwd(ptr(p), ps[0], ps[1], ps[2], ptr(t), ts[0], ts[1], ts[2], B, Cc, N, smooth, ptr(partial),
                           ptr(sums), ptr(loss), stream_ptr()), "n3d_dice_fwd")
    return loss, sums


def dice_bwd(p, t, smooth, sums, dloss, dp):
    B, Cc = p.shape[0], p.shape[1]
    N = p.shape[2] * p.shape[3] * p.shape[4]
    ps, ts, ds = _bcv_strides(p), _bcv_strides(t), _bcv_strides(dp)
    check(_lib.load().n3d_dice_bwd(ptr(p), ps[0], ps[1], ps[2], ptr(t), ts[0], ts[1], ts[2], B, Cc, N, smooth, ptr(sums),
                                   ptr(dloss), ptr(dp), ds[0], ds[1], ds[2], stream_ptr()), "n3d_dice_bwd")


class UpdateGuard:
    """what n3d_adam_step_guarded tests in front of an update (include/n3d.h, "Guarded update"): pointers (ints) to the device
    words {time-outs counted, time-outs acknowledged}, optionally the all-reduced peer flag (float), the loss scalar that
    becomes NaN when the update is withheld, and the host-visible word"""
    __slots__ = ("timeouts", "acked", "peer_flag", "loss", "host_word")

    def __init__(self, timeouts, acked, peer_flag=None, loss=None, host_word=None):
        self.timeouts, self.acked, self.peer_flag, self.loss, self.host_word = timeouts, acked, peer_flag, loss, host_word

    def with_loss(self, loss):
        """the same guard reporting into another loss scalar (tensor or None)"""
        return UpdateGuard(self.timeouts, self.acked, self.peer_flag, loss.data_ptr() if loss is not None else None, self.host_word)


def adam_step(param, grad, exp_avg, exp_avg_sq, step_t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0,
              grad_scale=1.0, inc_step=True, lr_dev=None, guard=None):
    """lr_dev: optional 1-element device tensor holding the learning rate (read by the kernel; survives graph replay).
    step_t: int32 step counter; a view made by `step_counter()` carries a ticket word behind it, and the update launch then
    counts the step itself (no second launch).  guard: UpdateGuard -- the update is withheld on the device when a stream
    hand-off of this step timed out (here or, data parallel, on a peer rank)"""
    inc = 0 if not inc_step else (2 if getattr(step_t, "_n3d_ticketed", False) else 1)
    if guard is None:
        check(_lib.load().n3d_adam_step(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, ptr(lr_dev),
                                        beta1, beta2, eps, weight_decay, grad_scale, ptr(step_t), inc, stream_ptr()),
              "n3d_adam_step")
        return
    vp = lambda v: C.c_void_p(v) if v else None
    check(_lib.load().n3d_adam_step_guarded(ptr(param), ptr(grad), ptr(exp_avg), ptr(exp_avg_sq), param.numel(), lr, ptr(lr_dev),
                                            beta1, beta2, eps, weight_decay, grad_scale, ptr(step_t), inc, vp(guard.timeouts), vp(guard.acked),
                                            vp(guard.peer_flag), vp(guard.loss), vp(guard.host_word), stream_ptr()),
          "n3d_adam_step_guarded")


def guard_flag(timeouts_ptr, acked_ptr, flag_ptr):
    """*flag = 1.0 if a hand-off of this rank timed out and is not acknowledged, else 0.0 (all-reduced with the gradients)"""
    check(_lib.load().n3d_guard_flag(C.c_void_p(timeouts_ptr), C.c_void_p(acked_ptr), C.c_void_p(flag_ptr), stream_ptr()), "n3d_guard_flag")


class HostWord:
    """one 32-bit word of pinned, device-mapped host memory (n3d_host_word_alloc): kernels store to it, `value` reads it without
    touching the HIP runtime"""

    def __init__(self):
        p = C.c_void_p()
        check(_lib.load().n3d_host_word_alloc(C.byref(p)), "n3d_host_word_alloc")
        self.ptr = p.value
        self._w = C.cast(p, C.POINTER(C.c_uint32))

    @property
    def value(self):
        return int(self._w[0])

    def clear(self):
        self._w[0] = 0

    def __del__(self):
        p, self.ptr = getattr(self, "ptr", None), None
        if p:
            try:
                _lib.load().n3d_host_word_free(C.c_void_p(p))
            except Exception:
                pass


# the side streams and their graphs, through libn3d's own HIP runtime (include/n3d.h, "stream hand-off")
def stream_create_low_priority():
    """-> HIP stream handle (int): non-blocking, lowest priority of the current device"""
    h = C.c_void_p()
    check(_lib.load().n3d_stream_create_low_priority(C.byref(h)), "n3d_stream_create_low_priority")
    return h.value


def stream_capture_begin(stream):
    check(_lib.load().n3d_stream_capture_begin(C.c_void_p(stream)), "n3d_stream_capture_begin")


def stream_capture_end(stream):
    """-> executable graph handle (int)"""
    ex = C.c_void_p()
    check(_lib.load().n3d_stream_capture_end(C.c_void_p(stream), C.byref(ex)), "n3d_stream_capture_end")
    return ex.value


def graph_launch(graph_exec, stream):
    check(_lib.load().n3d_graph_launch(C.c_void_p(graph_exec), C.c_void_p(stream)), "n3d_graph_launch")


def graph_destroy(graph_exec):
    _lib.load().n3d_graph_destroy(C.c_void_p(graph_exec))


def step_counter(device):
    """1-element int32 step counter (a view of two words: the second is the Adam kernel's ticket counter)"""
    two = torch.zeros(2, dtype=torch.int32, device=device)
    st = two[:1]
    st._n3d_ticketed = True
    return st


# ------------------------------------------------------------------------------------------ stream hand-off
def sync_signal(flag_ptr, step_ptr, bump=False):
    """publish *step to *flag behind everything on the current stream (include/n3d.h, "stream hand-off"); pointers are ints"""
    check(_lib.load().n3d_sync_signal(C.c_void_p(flag_ptr), C.c_void_p(step_ptr), 1 if bump else 0, stream_ptr()), "n3d_sync_signal")


SYNC_MAX_POLLS = [None]   # tests: force every wait of the side schedule to give up after this many polls


def sync_wait(flag_ptr, step_ptr, timeouts_ptr, bump=False, max_polls=5000000):
    """hold the current stream until *flag >= *step (bounded poll: ~0.3 us per try)"""
    if SYNC_MAX_POLLS[0] is not None:
        max_polls = SYNC_MAX_POLLS[0]
    check(_lib.load().n3d_sync_wait(C.c_void_p(flag_ptr), C.c_void_p(step_ptr), C.c_void_p(timeouts_ptr), 1 if bump else 0, int(max_polls),
                                    stream_ptr()), "n3d_sync_wait")


def sync_wait2(flag0_ptr, flag1_ptr, step_ptr, timeouts_ptr, bump=False, max_polls=5000000):
    """sync_wait on two flags with one launch"""
    if SYNC_MAX_POLLS[0] is not None:
        max_polls = SYNC_MAX_POLLS[0]
    check(_lib.load().n3d_sync_wait2(C.c_void_p(flag0_ptr), C.c_void_p(flag1_ptr), C.c_void_p(step_ptr), C.c_void_p(timeouts_ptr),
                                     1 if bump else 0, int(max_polls), stream_ptr()), "n3d_sync_wait2")


def stamp(ptr_):
    """diagnostic: the current stream stores the 100 MHz wall clock to the device uint64 at `ptr_` when it gets there"""
    check(_lib.load().n3d_stamp(C.c_void_p(ptr_), stream_ptr()), "n3d_stamp")


# ------------------------------------------------------------------------------------------ fused head
def dropout3d_gate(state, p, B, Cc):
    """(B, C) Dropout3d gate drawn on the device from `state` (uint32[3]: seed_lo, seed_hi, counter; the counter advances)"""
    gate = torch.empty((B, Cc), dtype=torch.float32, device=state.device)
    check(_lib.load().n3d_dropout3d_gate(ptr(state), float(p), B, Cc, ptr(gate), stream_ptr()), "n3d_dropout3d_gate")
    return gate


def _target_dtype(t):
    """N3D_F32 / N3D_U8 of a target tensor (None: no targets in the call)"""
    if t is None or t.dtype == torch.float32:
        return _lib.F32
    if t.dtype == torch.uint8:
        return _lib.U8
    raise N3DError(f"head: targets are float32 or uint8 ({{0, 1}} bytes), got {t.dtype}")


def _head_desc(x, w, bias, gate, dx=None, t=None):
    """x (and dx): View, or Planar -- the node-planar layout of include/n3d.h; t: the target tensor of the call (its storage type)"""
    td = _target_dtype(t)
    if isinstance(x, Planar):
        if dx is not None and not (isinstance(dx, Planar) and dx.cn == x.cn and dx.nn == x.nn):
            raise N3DError("head: a node-planar input needs a node-planar gradient of the same node layout")
        return _lib.Head(x.nodes[0].p.value, x.cn, x.dt, x.B, x.C, int(w.shape[0]), x.N, w.data_ptr(), bias.data_ptr(), _vp(gate),
                         x.node_stride, dx.node_stride if dx is not None else 0, x.cn, td)
    return _lib.Head(x.p.value, x.ld, x.dt, x.B, x.C, int(w.shape[0]), x.N, w.data_ptr(), bias.data_ptr(), _vp(gate), 0, 0, 0, td)


def head_fwd(x, w, bias, gate, t=None, smooth=1e-6, want_logits=False, want_p=True):
    """p = sigmoid(conv1x1x1(x * gate) + bias) as a contiguous (B, Co, D, H, W) tensor; with a target t also the Dice sums
    and loss from the same pass (t: float32, or uint8 bytes {0, 1}).  Returns (p, logits | None, sums | None, loss | None)."""
    lib = _lib.load()
    h = _head_desc(x, w, bias, gate, t=t)
    dev = x.t.device
    Co = int(w.shape[0])
    if not want_p and (t is None or want_logits):
        raise N3DError("head_fwd: the probabilities can only be skipped in the Dice mode without logits")
    p = torch.empty((x.B, Co, x.D, x.H, x.W), dtype=torch.float32, device=dev) if want_p else None
    logits = torch.empty_like(p) if want_logits else None
    sums = loss = partial = None
    ts = (0, 0, 0)
    if t is not None:
        ts = _bcv_strides(t)
        rows = int(lib.n3d_head_rows(x.N))
        partial = torch.empty((x.B, Co, rows, 3), dtype=torch.float64, device=dev)
        sums = torch.empty((x.B, Co, 3), dtype=torch.float64, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
    check(lib.n3d_head_fwd(C.byref(h), ptr(p), Co * x.N, x.N, 1, ptr(logits), ptr(t), ts[0], ts[1], ts[2], float(smooth), ptr(partial),
                           ptr(sums), ptr(loss), stream_ptr()), "n3d_head_fwd")
    return p, logits, sums, loss


def head_bwd(x, w, bias, gate, dx, dw, dbias, dp=None, t=None, sums=None, dloss=None, smooth=1e-6, accumulate=False):
    """backward of head_fwd in one pass: dx (+)=, dw, dbias.  Either dp (gradient w.r.t. p, any uniform strides) or
    (t, sums[, dloss]) for the fused Dice gradient.  x / dx: Views, or both Planar."""
    lib = _lib.load()
    h = _head_desc(x, w, bias, gate, dx if isinstance(x, Planar) else None, t=t)
    ws = None
    n = 0
    job = None
    if dw is not None or dbias is not None:
        n = int(lib.n3d_head_workspace_bytes(C.byref(h)))
        ws = torch.empty(max(n, 256), dtype=torch.uint8, device=x.t.device)
        job = FinalJob() if _ctx is not None else None
    ds = _bcv_strides(dp) if dp is not None else (0, 0, 0)
    ts = _bcv_strides(t) if t is not None else (0, 0, 0)
    check(lib.n3d_head_bwd(C.byref(h), ptr(dp), ds[0], ds[1], ds[2], ptr(t), ts[0], ts[1], ts[2], float(smooth), ptr(sums), ptr(dloss),
                           dx.nodes[0].p if isinstance(dx, Planar) else dx.p, dx.cn if isinstance(dx, Planar) else dx.ld, dx.dt,
                           ACCUMULATE if accumulate else 0, ptr(dw), ptr(dbias), ptr(ws), n,
                           C.byref(job) if job is not None else None, stream_ptr()), "n3d_head_bwd")
    if job is not None and job.nchunks > 0:
        _ctx.final.append(job)
        _ctx.keep.append(ws)


def ncdhw_to_ndhwc(src):
    """(B,C,D,H,W) contiguous NCDHW -> dense NDHWC (logical shape unchanged)."""
    B, Cc, D, H, W = src.shape
    dst = empty_ndhwc(B, Cc, D, H, W, src.device)
    check(_lib.load().n3d_ncdhw_to_ndhwc(ptr(src), ptr(dst), Cc, B, Cc, D * H * W, stream_ptr()), "n3d_ncdhw_to_ndhwc")
    return dst
