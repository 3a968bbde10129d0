"""Per-entry-point time table of one step, measured live (bench.py's `roofline_by_time`).

1. One EAGER step runs with every libn3d C-ABI call recorded (name + arguments).
2. Calls are grouped by (entry point, shape signature); each group's first call is replayed `reps` times inside a HIP graph
   and timed with HIP events on the launch stream -> microseconds per call without host launch cost (the same method as
   bench.conv_kernel_roofline).  The replay reuses the recorded pointers: the step's tensors have been returned to torch's
   caching allocator by then, but their memory stays mapped and nothing else runs, so the kernels do the same work on stale data.
3. Algorithmic FLOP and bytes per call come from the arguments (conv geometry, B / N / C of the epilogues; SURVEY 8(d):
   2*MACs, and "each tensor the op must read or write, once"), giving the fraction of the MFMA / HBM roofline per group.
The table is by C-ABI entry point + shape (one entry point = one kernel launch in a trainer step, where weights are
pre-packed and weight-gradient reductions deferred); profiles/r02_*_kernel_stats.csv hold the rocprofv3 per-kernel view of
the same commands.
"""
import ctypes as C
import collections

import torch

from nas_3d_unet_amd import _lib

PEAK_F32_TFLOPS = 157.3     # MI355X_MICROARCH.md: f32-input MFMA = vector rate
PEAK_BF16_TFLOPS = 2500.0   # dense bf16 MFMA
PEAK_HBM_GBS = 8000.0


# entry points that launch nothing (queries)
HOST_ONLY = {"n3d_conv_workspace_bytes", "n3d_conv_stats_rows", "n3d_conv_pack_info", "n3d_stats_rows", "n3d_fused_max_rows", "n3d_bwd_small2_ok",
             "n3d_head_rows", "n3d_head_workspace_bytes", "n3d_dice_rows", "n3d_last_error", "n3d_version", "n3d_device_ok", "n3d_comm_available",
             "n3d_dropout3d_uniform", "n3d_comm_unique_id", "n3d_comm_init", "n3d_comm_destroy"}


# entry points that are never replayed for timing: they change state the step depends on (weights and moments, the dropout
# generator) or hold a stream until another stream acts (the device-side hand-off); their coarse per-call event time is reported
NO_REPLAY = {"n3d_adam_step", "n3d_dropout3d_gate", "n3d_sync_wait", "n3d_sync_wait2", "n3d_sync_signal"}


# coefficient / gate kernels: a few workgroups over statistics rows (kilobytes) -- no roofline applies, their time is launch latency
LATENCY_ONLY = {"n3d_gn_coeffs2", "n3d_gn_coeffsN", "n3d_gn_bwd_coeffs", "n3d_gn_bwd_coeffs2", "n3d_gn_bwd_coeffsN", "n3d_node_fwd_coeffs",
                "n3d_node_bwd_coeffs", "n3d_se_gate_fwd", "n3d_se_gate_fwdN", "n3d_se_gate_bwd", "n3d_se_gate_bwdN", "n3d_plain_bwd_coeffs",
                "n3d_guard_flag", "n3d_stamp"}


class Recorder:
    """with Recorder() as r: ...   r.calls = [(name, args)] of every libn3d call made inside"""

    def __enter__(self):
        self.lib = _lib.load()
        self.calls = []
        self.orig = {}
        for name in _lib.PROTOTYPES:
            if name in HOST_ONLY:
                continue
            fn = getattr(self.lib, name)
            self.orig[name] = fn
            setattr(self.lib, name, self._wrap(name, fn))
        return self

    def _wrap(self, name, fn):
        def call(*args):
            self.calls.append((name, args))
            return fn(*args)
        return call

    def __exit__(self, *exc):
        for name, fn in self.orig.items():
            setattr(self.lib, name, fn)
        return False


def _val(a):
    if isinstance(a, C._SimpleCData):
        return a.value
    return a


def _geom(a):
    g = a._obj if hasattr(a, "_obj") else (a.contents if hasattr(a, "contents") else a)
    return g


def _gtuple(g):
    return tuple(getattr(g, f) for f, _ in _lib.ConvGeom._fields_)


def _esz(flag):
    return 2 if flag else 4


def _conv_cost(g, kind, flags):
    """(flop, bytes) of one conv call; kind: fwd | bwd_data | bwd_weight | bwd_both"""
    B, k3 = g.B, g.k ** 3
    No, Ni = g.Do * g.Ho * g.Wo, g.Di * g.Hi * g.Wi
    macs = B * No * g.Co * (1 if g.depthwise else g.Ci) * k3
    i_bytes, o_bytes = B * Ni * g.Ci, B * No * g.Co
    s16, d16 = bool(flags & _lib.SRC_BF16), bool(flags & _lib.DST_BF16)
    if kind == "fwd":
        return 2 * macs, i_bytes * _esz(s16) + o_bytes * _esz(d16)
    if kind == "fwdT":      # transposed forward: x on the o side, y on the i side
        return 2 * macs, o_bytes * _esz(s16) + i_bytes * _esz(d16)
    if kind == "bwd_data":  # dy (o side) -> dx (i side)
        return 2 * macs, o_bytes * _esz(s16) + i_bytes * _esz(d16)
    if kind == "bwd_dataT":
        return 2 * macs, i_bytes * _esz(s16) + o_bytes * _esz(d16)
    if kind == "bwd_weight":
        return 2 * macs, i_bytes * _esz(s16) + o_bytes * _esz(d16)
    return 4 * macs, (i_bytes + o_bytes) * 4 + i_bytes * 4   # bwd_both (fp32 only): x, dy, dx


def _struct(a):
    return a._obj if hasattr(a, "_obj") else a


def describe(name, args):
    """(signature, flop, bytes, dtype tag) of a recorded call, or (signature, None, None, '') when no cost model is attached"""
    v = [_val(a) for a in args]
    ew = lambda B, N, Cc, passes, bf: (0, passes * B * N * Cc * _esz(bf))
    try:
        if name in ("n3d_conv_fwd", "n3d_convT_fwd"):
            g = _geom(args[0]); fl = v[7]
            f, b = _conv_cost(g, "fwd" if name == "n3d_conv_fwd" else "fwdT", fl)
            return (name, _gtuple(g), fl & 0xC0), f, b
        if name in ("n3d_conv_bwd_data", "n3d_convT_bwd_data"):
            g = _geom(args[0]); fl = v[6]
            f, b = _conv_cost(g, "bwd_data" if name == "n3d_conv_bwd_data" else "bwd_dataT", fl)
            # what the op must also read, once each: the previous value of dx when it accumulates (the backward walk's second and third
            # writers of an activation's gradient) and the ReLU mask source of a conv with ReLU on load (the input itself)
            dx_bytes = g.B * (g.Di * g.Hi * g.Wi if name == "n3d_conv_bwd_data" else g.Do * g.Ho * g.Wo) * (g.Ci if name == "n3d_conv_bwd_data" else g.Co) * _esz(bool(fl & _lib.DST_BF16))
            extra = (dx_bytes if fl & _lib.ACCUMULATE else 0) + (dx_bytes if (name == "n3d_conv_bwd_data" and v[7]) else 0)
            return (name, _gtuple(g), fl & 0xC4, bool(name == "n3d_conv_bwd_data" and v[7])), f, b + extra
        if name in ("n3d_conv_bwd_weight", "n3d_convT_bwd_weight"):
            g = _geom(args[0]); fl = v[7]
            f, b = _conv_cost(g, "bwd_weight", fl)
            return (name, _gtuple(g), fl & 0xC0), f, b
        if name in ("n3d_conv_bwd_both", "n3d_convT_bwd_both"):
            g = _geom(args[0])
            f, b = _conv_cost(g, "bwd_both", 0)
            return (name, _gtuple(g)), f, b
        if name in ("n3d_conv_fwd2", "n3d_conv_bwd_both2", "n3d_conv_bwd_data2"):
            sig, f, b = [name], 0, 0
            for c in (_struct(args[0]), _struct(args[1])):
                g = c.g.contents
                if name == "n3d_conv_fwd2":
                    ff, bb = _conv_cost(g, "fwdT" if c.transposed else "fwd", c.flags)
                elif name == "n3d_conv_bwd_both2":
                    ff, bb = _conv_cost(g, "bwd_both", 0)
                else:
                    ff, bb = _conv_cost(g, "bwd_data", 0)
                sig.append(_gtuple(g)); f += ff; b += bb
            return tuple(sig), f, b
        if name == "n3d_conv_fwdN":
            calls, n = args[0], v[1]
            sig, f, b = [name], 0, 0
            for i in range(n):
                c = calls[i]
                g = c.g.contents
                ff, bb = _conv_cost(g, "fwdT" if c.transposed else "fwd", c.flags)
                sig.append(_gtuple(g)); f += ff; b += bb
            return tuple(sig), f, b
        if name == "n3d_dwconv_batch":         # depthwise gather passes of up to 8 primitives, one launch
            jobs, n = args[0], v[1]
            sig, f, b = [name], 0, 0
            for i in range(n):
                j = jobs[i]
                g = j.g.contents
                ff, bb = _conv_cost(g, "bwd_data" if j.data_grad else "fwd", 0)
                sig.append(_gtuple(g) + (int(j.data_grad),)); f += ff; b += bb + (bb // 2 if (j.flags & _lib.ACCUMULATE) else 0)
            return tuple(sig), f, b
        if name == "n3d_affine_actN":          # out (+)= sum of n normalised terms: n raw reads, one write (+ one read when accumulating)
            n, B, N, Cc, fl = v[1], v[4], v[5], v[6], v[7]
            return (name, n, B, N, Cc, fl & 0x44), *ew(B, N, Cc, n + (2 if fl & 4 else 1), False)
        if name == "n3d_affine_act_bwd_reduceN":   # the node gradient once, every raw once (sums are rows of doubles: negligible)
            n, B, N, Cc = v[3], v[4], v[5], v[6]
            return (name, n, B, N, Cc), *ew(B, N, Cc, n + 1, False)
        if name == "n3d_affine_act_bwd_applyN":    # ... and every d(raw) written
            n, B, N, Cc = v[3], v[4], v[5], v[6]
            return (name, n, B, N, Cc), *ew(B, N, Cc, 2 * n + 1, False)
        if name == "n3d_affine_act_bwd_apply_sum":  # terms with the same target are summed in registers: one write (+ read) per TARGET
            terms, n, B, N, Cc = args[2], v[3], v[4], v[5], v[6]
            tg = {}
            for i in range(n):
                t = terms[i]
                tg.setdefault(t.draw, bool(t.pad_))     # (pad_ carries "accumulate" of the first term of a target)
            passes = n + 1 + sum(2 if acc else 1 for acc in tg.values())
            return (name, n, len(tg), B, N, Cc), *ew(B, N, Cc, passes, False)
        if name in LATENCY_ONLY:
            scal = tuple(x for x in v if isinstance(x, (int, float)) and not (isinstance(x, int) and x > (1 << 32)))
            return (name,) + scal[:6], 0, 0
        if name == "n3d_affine_act_gn2":
            B, N, Cc, fl = v[8], v[9], v[10], v[11]
            return (name, B, N, Cc, fl & 0x44, bool(v[6])), *ew(B, N, Cc, 4 if v[6] else 3, fl & _lib.ACT_BF16)
        if name == "n3d_affine_act2":
            B, N, Cc, fl = v[6], v[7], v[8], v[9]
            return (name, B, N, Cc, fl & 0x44, bool(v[4])), *ew(B, N, Cc, 4 if v[4] else 3, fl & _lib.ACT_BF16)
        if name in ("n3d_affine_act_bwd_reduce2", "n3d_affine_act_bwd_apply_gn2", "n3d_affine_act_bwd_apply2", "n3d_affine_act_bwd_small2"):
            B, N, Cc = v[6], v[7], v[8]
            bf = _struct(args[4]).dtype == _lib.BF16
            two = bool(v[2])
            passes = (4 if two else 3) if name == "n3d_affine_act_bwd_reduce2" else (6 if two else 5)
            return (name, B, N, Cc, bf, two), *ew(B, N, Cc, passes, bf)
        if name == "n3d_affine_act":
            B, N, Cc, fl = v[7], v[8], v[9], v[10]
            return (name, B, N, Cc, fl & 0x46), *ew(B, N, Cc, 3 if fl & 4 else 2, fl & _lib.ACT_BF16)
        if name == "n3d_affine_act_gn":
            B, N, Cc, fl = v[11], v[12], v[13], v[14]
            return (name, B, N, Cc, fl & 0x46), *ew(B, N, Cc, 3 if fl & 4 else 2, fl & _lib.ACT_BF16)
        if name == "n3d_affine_act_bwd_reduce":
            B, N, Cc, fl = v[6], v[7], v[8], v[9]
            return (name, B, N, Cc, fl & 0x42), *ew(B, N, Cc, 2, fl & _lib.ACT_BF16)
        if name == "n3d_affine_act_bwd_apply":
            B, N, Cc, fl = v[11], v[12], v[13], v[14]
            return (name, B, N, Cc, fl & 0x46), *ew(B, N, Cc, 4 if fl & 4 else 3, fl & _lib.ACT_BF16)
        if name == "n3d_affine_act_bwd_apply_gn":
            B, N, Cc, fl = v[14], v[15], v[16], v[18]
            return (name, B, N, Cc, fl & 0x46), *ew(B, N, Cc, 4 if fl & 4 else 3, fl & _lib.ACT_BF16)
        if name == "n3d_channel_stats_t":
            return (name, v[2], v[3], v[4], v[5]), *ew(v[3], v[4], v[5], 1, v[2] == _lib.BF16)
        if name in ("n3d_head_fwd", "n3d_head_bwd"):
            h = _struct(args[0])
            xb = h.B * h.N * h.Ci * _esz(h.x_dtype == _lib.BF16)
            pb = h.B * h.N * h.Co * 4
            if name == "n3d_head_fwd":   # reads x (+ t), writes p
                return (name, h.B, h.N, h.Ci, h.Co, h.x_dtype), 2 * h.B * h.N * h.Ci * h.Co, xb + pb * (2 if v[6] else 1)
            return (name, h.B, h.N, h.Ci, h.Co, h.x_dtype), 6 * h.B * h.N * h.Ci * h.Co, 2 * xb + pb
        if name == "n3d_adam_step":
            return (name, v[4]), 0, 7 * 4 * v[4]
        if name == "n3d_pack_batch":      # reads every weight once per job, writes its packed form (layout 0 pads the channels)
            jobs, n, byts = args[0], v[1], 0
            for i in range(n):
                j = jobs[i]
                cs, cd = (j.Co, j.Ci) if j.data_grad else (j.Ci, j.Co)
                el = j.taps * (cs * j.cdp if j.layout == 0 else (cs * cd if j.layout == 1 else j.Co * j.Co))
                byts += 4 * j.Co * j.Ci * j.taps + (2 if j.layout >= 4 else 4) * el
            return (name, n), 0, byts
        if name == "n3d_wgrad_finalize_batch":   # reads every partial slab, writes each gradient once
            jobs, n, byts = args[0], v[1], 0
            for i in range(n):
                j = jobs[i]
                byts += 4 * (j.nchunks * (j.ntiles * j.ci_t * j.co_t + j.tco * j.co_t) + j.Co * j.Ci * j.taps + j.Co)
            return (name, n), 0, byts
    except Exception as e:      # a cost model that does not fit the recorded arguments must not cost the table -- but it must show
        describe.errors[name] = "%s: %s" % (type(e).__name__, e)
    scal = tuple(x for x in v if isinstance(x, (int, float)) and not (isinstance(x, int) and x > (1 << 32)))
    return (name,) + scal[:6], None, None


describe.errors = {}


def _time_call(name, args, device, reps=20, rounds=3):
    lib = _lib.load()
    fn = getattr(lib, name)
    from nas_3d_unet_amd.train import capture_stream
    side = capture_stream(device)      # the process-wide capture stream (every extra stream is another hardware queue)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        a = list(args[:-1]) + [sp]
        _lib.check(fn(*a), name)           # once eagerly on the side stream
        torch.cuda.synchronize()
        # raw capture calls: the torch.cuda.graph context manager empties the allocator's cache first, which would unmap the
        # recorded step's (freed) tensors that the replay still addresses
        graph.capture_begin(capture_error_mode="thread_local")
        try:
            for _ in range(reps):
                _lib.check(fn(*a), name)
        finally:
            graph.capture_end()
    stream = torch.cuda.current_stream()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(rounds):
        graph.replay()
    e1.record(stream)
    e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)   # us per call


def table(run_step, device, top=5, candidates=14):
    """run_step(): one EAGER step of the workload (called twice: warm-up, then recorded).
    Returns (rows for the JSON line, number of C-ABI calls per step)."""
    run_step()
    torch.cuda.synchronize()
    with Recorder() as rec:
        # per-call HIP events on the launch stream rank the groups (coarse: includes launch gaps); the top groups are re-timed
        evs = []
        orig_wrap = rec._wrap

        def timed(name, fn):
            def call(*args):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = fn(*args)
                e1.record()
                rec.calls.append((name, args))
                evs.append((e0, e1))
                return r
            return call
        for name, fn in rec.orig.items():
            setattr(rec.lib, name, timed(name, fn))
        run_step()
        torch.cuda.synchronize()
    groups = collections.OrderedDict()
    for (name, args), (e0, e1) in zip(rec.calls, evs):
        sig, flop, byts = describe(name, args)
        gr = groups.setdefault(sig, dict(name=name, args=args, calls=0, coarse_us=0.0, flop=flop, bytes=byts))
        gr["calls"] += 1
        gr["coarse_us"] += e0.elapsed_time(e1) * 1e3
    # (entry points that are never replayed only have a coarse per-call event time, host gaps included: they do not compete for the table)
    ranked = sorted(((sig, gr) for sig, gr in groups.items() if gr["name"] not in NO_REPLAY), key=lambda kv: -kv[1]["coarse_us"])[:candidates]
    rows = []
    for sig, gr in ranked:
        if gr["name"].startswith("n3d_comm") or gr["name"] in NO_REPLAY:
            us = gr["coarse_us"] / gr["calls"]
        else:
            try:
                us = _time_call(gr["name"], gr["args"], device)
            except Exception:
                us = gr["coarse_us"] / gr["calls"]
        row = {"entry": gr["name"], "shape": _shape_text(sig), "calls_per_step": gr["calls"], "us_per_call": round(us, 2),
               "us_per_step": round(us * gr["calls"], 1)}
        if gr["name"] in LATENCY_ONLY:
            row.update({"bound": "latency", "frac": None, "note": "coefficient kernel over statistics rows (kilobytes): launch latency, no roofline"})
        elif gr["flop"] is None:
            row.update({"bound": None, "frac": None, "note": "no cost model" + ((": " + describe.errors[gr["name"]]) if gr["name"] in describe.errors else "")})
        else:
            bf16_mfma = gr["name"] in ("n3d_conv_fwd", "n3d_conv_bwd_data") and (sig[2] & 0xC0) == 0xC0 and sig[1][9] == 3 and sig[1][10] == 1
            peak = PEAK_BF16_TFLOPS if bf16_mfma else PEAK_F32_TFLOPS   # the bf16-storage 3x3x3 kernels run v_mfma_f32_4x4x4_16b_bf16
            row["mfma_peak_tflops"] = peak
            t_mfma = gr["flop"] / (peak * 1e12) * 1e6
            t_hbm = gr["bytes"] / (PEAK_HBM_GBS * 1e9) * 1e6
            row.update({"algorithmic_flop": gr["flop"], "algorithmic_bytes": gr["bytes"], "bound": "mfma" if t_mfma > t_hbm else "hbm",
                        "achieved_tflops": round(gr["flop"] / us / 1e6, 2), "achieved_gbs": round(gr["bytes"] / us / 1e3, 1),
                        "frac": round(max(t_mfma, t_hbm) / us, 4)})
        rows.append(row)
    rows.sort(key=lambda r: -r["us_per_step"])
    return rows[:top], len(rec.calls)


def _shape_text(sig):
    out = []
    for x in sig[1:]:
        if isinstance(x, tuple) and len(x) == 14:
            B, Di, Hi, Wi, Ci, Do, Ho, Wo, Co, k, st, dil, pad, dw = x
            out.append("B%d %dx%dx%dx%d->%dx%dx%dx%d k%d s%d d%d%s" % (B, Ci, Di, Hi, Wi, Co, Do, Ho, Wo, k, st, dil, " dw" if dw else ""))
        else:
            out.append(str(x))
    return " ".join(out)
