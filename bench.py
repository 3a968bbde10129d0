#!/usr/bin/env python3
"""Headline benchmark: 4x64^3 patches/s of one searched-net (G_conv) train step
(zero_grad -> forward -> Dice -> backward -> Adam, + RCCL gradient all-reduce when N > 1),
batch 2 per GPU, fp32, synthetic data, random-init weights  (BASELINE.json configs[1] / [3]).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

Prints ONE JSON line on rank 0 with the contract fields plus
  "roofline":     the dominant kernel (3x3x3 conv, C=4 @ 64^3) timed live with HIP events
  "cpu_baseline": the CPU oracle (oracle/, torch CPU = the reference's own arithmetic) on the host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# benchmark genotype G_conv (the reference ships no genotype)
G_CONV = dict(
    down=[("down_conv", 0), ("down_dil_conv", 1), ("down_conv", 1), ("conv", 2), ("dil_conv", 2), ("conv", 3)],
    up=[("conv", 0), ("up_conv", 1), ("up_conv", 1), ("dil_conv", 2), ("conv", 3), ("up_dil_conv", 1)],
)
CFG = dict(in_channels=4, init_n_kernels=4, out_channels=3, depth=4, n_nodes=3, channel_change=True)
# algorithmic work per 4x64^3 patch, searched net / G_conv
FLOP_FWD_BWD_PER_PATCH = 5.41e9
BYTES_FWD_PER_PATCH = 148.3e6
PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32-input MFMA = vector rate
PEAK_HBM_GBS = 8000.0


def synthetic_batch(batch, size, seed):
    """Masked clipped-Gaussian volume in [10,110] inside a centred ball, nested-ball targets."""
    rng = np.random.default_rng(seed)
    g = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    r = np.sqrt(g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2)
    x = np.clip(50.0 + 25.0 * rng.standard_normal((batch, 4, size, size, size)), 10.0, 110.0) * (r <= 0.45 * size)
    t = np.stack([(r <= 0.22 * size), (r <= 0.30 * size), (r <= 0.12 * size)]).astype(np.float32)
    return x.astype(np.float32), np.broadcast_to(t, (batch,) + t.shape).copy()


def to_patch_layout(x):
    """(B, C, D, H, W) logical shape in NDHWC storage -- what the data step (datastep.patch_batch / n3d_patch_batch) hands the
    net, so the step does not start with a layout conversion"""
    return x.contiguous(memory_format=torch.channels_last_3d)


def conv_kernel_roofline(device, batch, size, iters=40, bf16=False, c=4):
    """Time the dominant FLOP kernel -- the 3x3x3 stride-1 conv at C=4 on (batch, 4, size^3), the shape of up-cell 4
    (43 % of the net's FLOPs; the same kernel serves its data gradient) -- with HIP events on the launch stream.
    The launches are replayed from a HIP graph with pre-packed weights, so the figure is kernel time plus the
    ~1.7 us dependent-launch boundary, not host launch cost.  Algorithmic FLOPs = 2*B*S^3*C*C*27 per launch,
    algorithmic bytes = input + output tensor.
    bf16 (BASELINE configs[4]): the same conv on bf16-stored tensors (conv_vox64b_kernel, v_mfma_f32_4x4x4_16b_bf16), priced
    against HBM: bytes / time / 8 TB/s."""
    from nas_3d_unet_amd import kernels as K
    dt = torch.bfloat16 if bf16 else torch.float32
    x = K.as_view(K.empty_ndhwc(batch, c, size, size, size, device, dt).normal_())
    y = K.as_view(K.empty_ndhwc(batch, c, size, size, size, device, dt))
    w = torch.randn(c, c, 3, 3, 3, device=device) * 0.1
    b = torch.randn(c, device=device) * 0.1
    g = K.conv_geom(batch, size, size, size, c, c, 3, 1, 1, 1)
    rows = K.conv_stats_rows(g, False)
    stats = torch.empty((batch, rows, c, 2), dtype=torch.float64, device=device)
    ctx = K.StepContext(device)
    with K.step_context(ctx):
        K.conv_fwd(g, x, w, b, y, 0, None, stats, False)   # records the pack job
        ctx.freeze()
        ctx.pack_all()
        K.conv_fwd(g, x, w, b, y, 0, None, stats, False)   # pre-packed path
        torch.cuda.synchronize()
        from nas_3d_unet_amd.train import capture_stream
        side = capture_stream(device)     # the process-wide capture stream: every extra stream is another hardware queue
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, capture_error_mode="thread_local"):
                for _ in range(iters):
                    K.conv_fwd(g, x, w, b, y, 0, None, stats, False)
    stream = torch.cuda.current_stream()
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 5
    e0.record(stream)
    for _ in range(reps):
        graph.replay()
    e1.record(stream)
    e1.synchronize()
    sec = e0.elapsed_time(e1) * 1e-3 / (iters * reps)
    flops = 2.0 * batch * size ** 3 * c * c * 27
    bytes_ = 2.0 * batch * size ** 3 * c * (2 if bf16 else 4)
    ach = flops / sec / 1e12
    # HBM traffic per launch: PMC counters (FETCH_SIZE x 2 on gfx950 + WRITE_SIZE) need their own rocprofv3 --pmc passes, so the
    # figure is read from the tracked summary of that pass (tools/collect_pmc_r05.sh, run by tools/collect_r06.sh), not measured inside this run
    traffic, src = None, None
    for rnd in ("r06", "r05", "r04", "r03", "r02"):
        pmc = os.path.join(ROOT, "profiles", "%s_pmc_conv_vox64%s_%s_2x%dx%d.json" % (rnd, "b" if bf16 else "", "bf16" if bf16 else "f32", c, size))
        if os.path.exists(pmc):
            try:
                traffic, src = json.load(open(pmc)).get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(pmc)
            except Exception:
                traffic = None
            break
    common = {"us_per_launch": round(sec * 1e6, 2), "algorithmic_flop_per_launch": flops, "algorithmic_bytes_per_launch": bytes_,
              "traffic": traffic, "traffic_source": (src + " (separate rocprofv3 --pmc passes of the same launch, not this run)") if src else None}
    if bf16:
        gbs = bytes_ / sec / 1e9
        return {"bound": "hbm", "kernel": "conv_vox64b_kernel<%d,...>: 3x3x3 s1 d1 conv, C=%d, bf16 storage, (%d,%d,%d^3), GN-stats epilogue" % (c, c, batch, c, size),
                "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": round(gbs / PEAK_HBM_GBS, 4),
                "achieved_tflops": round(ach, 3), **common}
    return {"bound": "mfma", "kernel": "conv_vox64_kernel<%d,...>: 3x3x3 s1 d1 conv, C=%d, (%d,%d,%d^3), GN-stats epilogue" % (c, c, batch, c, size),
            "achieved": round(ach, 3), "peak": PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": round(ach / PEAK_FP32_TFLOPS, 4),
            "algorithmic_gbs": round(bytes_ / sec / 1e9, 1), **common}


def usable_cores():
    """cores this process may really use: affinity mask capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(per))))
    except Exception:
        pass
    return max(1, n)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(batch, size, budget_s=22.0):
    """The CPU oracle (functional torch-CPU restatement, proven equal to the reference by the golden
    tests) timed on this box's host cores: forward + Dice + backward at the same batch, once with every usable core and once
    with ONE thread, both on a bounded sample."""
    from oracle import ref_path as orc
    cores = usable_cores()
    P = orc.make_params(orc.searched_param_specs(orc.DEFAULT_CFG, orc.G_CONV), requires_grad=True)
    xn, tn = synthetic_batch(batch, size, 99)
    x, t = torch.from_numpy(xn), torch.from_numpy(tn)

    def one():
        for q in P.values():
            q.grad = None
        l = orc.dice_loss(orc.searched_forward(P, x, orc.G_CONV), t)
        l.backward()

    def timed(threads, max_iters, budget):
        torch.set_num_threads(threads)
        one()  # warm-up
        times, t_all = [], time.perf_counter()
        while len(times) < max_iters and (time.perf_counter() - t_all) < budget:
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
        return float(np.median(times)), len(times)

    # best of a small thread sweep (round-2 review: "all cores" was slower than one thread on a contended box)
    sweep = sorted({1, min(4, cores), min(8, cores), cores})
    share = budget_s / len(sweep)
    runs = {}
    for th in sweep:
        runs[th] = timed(th, 7 if th > 1 else 3, share)
    torch.set_num_threads(cores)
    best = min(runs, key=lambda th: runs[th][0])
    med, n = runs[best]
    return {"value": round(batch / med, 3), "unit": "patches/s", "cores": best, "kind": "port", "cpu_model": cpu_model(), "usable_cores": cores,
            "thread_sweep": {str(th): {"value": round(batch / m, 3), "iterations": k, "median_s": round(m, 3)} for th, (m, k) in runs.items()},
            "one_thread": {"value": round(batch / runs[1][0], 3), "unit": "patches/s", "cores": 1,
                           "sample": "%d timed fwd+bwd iterations, median %.2f s" % (runs[1][1], runs[1][0])},
            "sample": "best of threads in %s: %d timed fwd+bwd iterations of the same net at batch %d, 4x%d^3 fp32 with %d threads (median %.3f s)"
                      % (sweep, n, batch, size, best, med)}


def search_step_bench(args, device):
    """BASELINE configs[2]: nas.py supernet search step (search.py:211-238): architecture pass on a validation batch
    (Adam on the alphas) + weight pass on a training batch (Adam on the kernel weights), batch 2 each, one GPU."""
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    torch.manual_seed(1234)
    net = nas.ShellNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"], CFG["n_nodes"], False,
                       CFG["channel_change"]).to(device)
    net.train()
    tr = SearchTrainer(net, graph=not args.no_graph, comm=args.comm)
    xn, tn = synthetic_batch(args.batch, args.size, 1234)
    vxn, vtn = synthetic_batch(args.batch, args.size, 4321)
    x, t, vx, vt = (torch.from_numpy(a).to(device) for a in (xn, tn, vxn, vtn))
    x, vx = to_patch_layout(x), to_patch_layout(vx)
    for _ in range(args.warmup):
        tr.step(x, t, vx, vt)
    if not args.no_graph and args.warmup > 0:
        own = tr.input_buffers()              # as in main(): the batches sit in the trainer's input buffers
        for dst, src in zip(own, (x, t, vx, vt)):
            dst.copy_(src)
        x, t, vx, vt = own
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        la, lw = tr.step(x, t, vx, vt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    tr.check_sync()
    extra = {"sync_timeouts": tr.sync_timeouts(), "schedule": "three streams" if tr._use_side else "single stream"}
    if not args.no_kernel_table:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        try:
            import kernel_table
            side, tr.side = tr.side, None      # the single-stream schedule: one entry point = one launch of the dependent chain
            try:
                rows, ncalls = kernel_table.table(lambda: tr._both(x, t, vx, vt), device)
            finally:
                tr.side = side
            extra.update({"roofline_by_time": rows, "launches_per_step": ncalls})
        except Exception as e:
            extra["roofline_by_time"] = "failed: %s" % (str(e)[:200],)
    print(json.dumps({
        "metric": "supernet search steps/sec (arch pass + weight pass, each fwd + Dice + bwd + Adam)", "value": round(args.steps / dt, 3),
        "unit": "steps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "patches_per_s": round(2 * args.batch * args.steps / dt, 2),
        "config": {"workload": "nas.py ShellNet search step, train batch=%d + val batch=%d 4x%d^3 fp32" % (args.batch, args.batch, args.size),
                   "hip_graph": not args.no_graph, "input_layout": "NDHWC (as n3d_patch_batch emits)", "loss_arch": round(float(la), 5), "loss_weight": round(float(lw), 5)}, **extra}), flush=True)


def other_configs(device, batch, no_graph, targets="f32"):
    """The other single-GPU BASELINE configs, timed in the same process after the headline measurement (10 steps each after 3 warm-up
    steps, HIP-graph replay) so that the driver's own run of the default command carries them: configs[2] (supernet search step at
    4x64^3), the 4x128^3 train step in fp32 and configs[4] (4x128^3 with bf16 storage).  Each entry is a measurement or an error string."""
    from nas_3d_unet_amd import nas, searched
    from nas_3d_unet_amd.train import SearchTrainer, Trainer
    out = {}

    def timed(tr, bufs, steps=10, warmup=3):
        for _ in range(warmup):
            tr.step(*bufs)
        if not no_graph:
            own = tr.input_buffers()          # as in main(): the batch sits in the trainer's input buffers
            for dst, src in zip(own, bufs):
                dst.copy_(src)
            bufs = own
        step = lambda: tr.step(*bufs)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        sec = (time.perf_counter() - t0) / steps
        tr.check_sync()        # a timed-out hand-off (update withheld on the device) must not be reported as a measurement
        return sec

    def search():
        torch.manual_seed(1234)
        net = nas.ShellNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"], CFG["n_nodes"], False, CFG["channel_change"]).to(device)
        net.train()
        tr = SearchTrainer(net, graph=not no_graph)
        xn, tn = synthetic_batch(batch, 64, 1234)
        vxn, vtn = synthetic_batch(batch, 64, 4321)
        x, t, vx, vt = (torch.from_numpy(a).to(device) for a in (xn, tn, vxn, vtn))
        x, vx = to_patch_layout(x), to_patch_layout(vx)
        sec = timed(tr, (x, t, vx, vt))
        return {"ms_per_step": round(sec * 1e3, 3), "steps_per_s": round(1.0 / sec, 3), "patches_per_s": round(2 * batch / sec, 2),
                "schedule_ms": [round(v * 1e3, 3) for v in tr.schedule_times] if getattr(tr, "schedule_times", None) else None}

    def train128(storage):
        torch.manual_seed(1234)
        net = searched.SearchedNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"], CFG["n_nodes"], CFG["channel_change"],
                                   searched.Genotype(**G_CONV)).to(device)
        net.train()
        tr = Trainer(net, graph=not no_graph, storage=storage)
        xn, tn = synthetic_batch(batch, 128, 1234)
        x, t = to_patch_layout(torch.from_numpy(xn).to(device)), torch.from_numpy(tn).to(device)
        if targets == "u8":
            t = t.to(torch.uint8)
        sec = timed(tr, (x, t))
        return {"ms_per_step": round(sec * 1e3, 3), "patches_per_s": round(batch / sec, 2),
                "schedule_ms": [round(v * 1e3, 3) for v in tr.schedule_times] if getattr(tr, "schedule_times", None) else None}

    for name, fn in (("configs[2]: nas.py supernet search step, batch 2 + 2, 4x64^3 fp32", search),
                     ("searched.py train step, batch 2, 4x128^3 fp32", lambda: train128(None)),
                     ("configs[4]: searched.py train step, batch 2, 4x128^3 bf16 storage", lambda: train128("bf16"))):
        try:
            out[name] = fn()
        except Exception as e:   # the headline line must not depend on these
            out[name] = "failed: %s" % (str(e)[:200],)
        torch.cuda.empty_cache()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=2, help="patches per GPU")
    ap.add_argument("--size", type=int, default=64)
    ap.add_argument("--dtype", choices=["f32", "bf16"], default="f32",
                    help="f32: the reference's arithmetic and storage (the contract default); bf16: bf16 STORAGE of the HBM-bound levels' "
                         "activations, fp32 arithmetic (BASELINE configs[4], quoted at --size 128)")
    ap.add_argument("--targets", choices=["f32", "u8"], default="f32",
                    help="storage of the three target maps in HBM: f32 (the reference's cast, train.py:118; the contract default) or u8 "
                         "(the generator's booleans as bytes, n3d_patch_batch's N3D_PATCH_T_U8: same losses bit for bit)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-kernel-table", action="store_true", help="skip roofline_by_time (the per-entry-point time table of one step)")
    ap.add_argument("--no-other-configs", action="store_true", help="skip other_configs (search step and the 128^3 train steps, timed after the headline run)")
    ap.add_argument("--buckets", type=int, default=None, help="gradient buckets of the data-parallel exchange (default: N3D_DP_BUCKETS or 1); "
                    ">= 2: all-reduce of a bucket on a side stream under the backward of the next one")
    ap.add_argument("--comm", choices=["torch", "rccl"], default=None, help="gradient all-reduce through the C ABI's n3d_comm_* (default) or torch.distributed")
    ap.add_argument("--workload", choices=["train", "search"], default="train",
                    help="train: searched-net train step (BASELINE configs[1], the contract default); "
                         "search: supernet search step, arch pass + weight pass (configs[2]; informational, N=1 only)")
    args = ap.parse_args()

    if args.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: start the N ranks as fresh child processes (one per GPU) BEFORE anything here touches the GPU,
        # and leave with their exit code; the JSON line is rank 0's
        import subprocess
        n_dev = torch.cuda.device_count()      # counting devices does not initialise the GPU
        if n_dev < args.gpus:
            sys.exit("bench.py: --gpus %d requested, but this node shows %d GPU(s)" % (args.gpus, n_dev))
        port = os.environ.get("MASTER_PORT", "29511")
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
               "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd, env=env))
    if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
        sys.exit("bench.py: --gpus %d does not match WORLD_SIZE=%s of the launcher" % (args.gpus, os.environ["WORLD_SIZE"]))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 or os.environ.get("N3D_FORCE_DP") == "1":   # (N3D_FORCE_DP: 1-rank RCCL group, exercises the N > 1 code path)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", rank=rank, world_size=world)
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)

    from nas_3d_unet_amd import _lib, searched
    from nas_3d_unet_amd.train import Trainer, reserve_side_streams
    _lib.require_device()
    reserve_side_streams(device)     # before anything else runs on the GPU: the side streams of every trainer this process builds
    if args.workload == "search":
        return search_step_bench(args, device)

    torch.manual_seed(1234)  # same random-init weights on every rank (then broadcast anyway)
    net = searched.SearchedNet(CFG["in_channels"], CFG["init_n_kernels"], CFG["out_channels"], CFG["depth"],
                               CFG["n_nodes"], CFG["channel_change"], searched.Genotype(**G_CONV)).to(device)
    net.train()  # head Dropout3d(0.5) active, as in training (searched.py:91-93)
    trainer = Trainer(net, graph=not args.no_graph, n_buckets=args.buckets, comm=args.comm, storage="bf16" if args.dtype == "bf16" else None)

    xn, tn = synthetic_batch(args.batch, args.size, 1234 + rank)
    x, t = to_patch_layout(torch.from_numpy(xn).to(device)), torch.from_numpy(tn).to(device)
    if args.targets == "u8":
        t = t.to(torch.uint8)

    for _ in range(args.warmup):
        loss = trainer.step(x, t)
    if not args.no_graph and args.warmup > 0:
        # the synthetic batch lives in the trainer's own input buffers, where the on-device data step (n3d_patch_batch) would
        # put it: resident in HBM when the timed region starts, no per-step staging copy
        bx, bt = trainer.input_buffers()
        bx.copy_(x); bt.copy_(t)
        x, t = bx, bt
    # the ranks meet through the trainers' own RCCL communicator (nas_3d_unet_amd.comm: ONE per process; torch.distributed only
    # carries the rendezvous -- its NCCL backend never builds a communicator, i.e. no second set of RCCL queues next to the step's)
    from nas_3d_unet_amd import comm as n3d_comm
    cm = n3d_comm.for_group(None, device) if world > 1 else None
    if cm is not None:
        cm.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = trainer.step(x, t)
    if cm is not None:
        cm.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if cm is not None:
        dt = max(cm.gather_floats(dt))       # MAX over ranks
    final_loss = float(loss)
    # the side-stream schedule orders its streams with BOUNDED device-side waits; a wait that gave up withholds the update on the
    # device (n3d_adam_step_guarded) -- the line certifies that none did during the steps it reports (summed over the ranks)
    sync_timeouts = trainer.sync_timeouts()
    if cm is not None:
        sync_timeouts = int(sum(cm.gather_floats(sync_timeouts)))
    trainer.check_sync()     # raises (no line) if a hand-off timed out

    if rank == 0:
        patches = world * args.batch * args.steps
        value = patches / dt
        out = {
            "metric": "4x%d^3 patches/sec (train step: fwd + Dice + bwd + Adam%s)" % (args.size, " + RCCL all-reduce" if world > 1 else ""),
            "value": round(value, 2), "unit": "patches/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "sync_timeouts": sync_timeouts,
            "vs_baseline": None, "dtype": "f32" if args.dtype == "f32" else "bf16 storage (levels with <= 8 channels per node, stems, head input) / f32 arithmetic", "data": "synthetic",
            "config": {"workload": "searched.py SearchedNet/G_conv train step, batch=%d 4x%d^3 %s per GPU" % (args.batch, args.size, "fp32" if args.dtype == "f32" else "bf16-storage"),
                       "global_batch": world * args.batch, "patch": [4, args.size, args.size, args.size],
                       "parallelism": "dp%d" % world, "dp_buckets": len(trainer.sync.ranges) if trainer.dp_path else None, "hip_graph": not args.no_graph, "input_layout": "NDHWC (as n3d_patch_batch emits)", "targets": args.targets, "final_loss": round(final_loss, 5),
                       "schedule": ("side-stream weight gradients" if getattr(trainer, "_use_side", False) else "single stream"),
                       "schedule_ms": [round(v * 1e3, 3) for v in trainer.schedule_times] if getattr(trainer, "schedule_times", None) else None},
            # algorithmic work per patch scales with the voxel count (figures quoted at 64^3); bf16 storage halves the
            # bytes of the levels it covers (~94 % of the activation bytes)
            "whole_net": (lambda vox, byt: {"tflops_fwd_bwd": round(value * FLOP_FWD_BWD_PER_PATCH * vox / 1e12, 3),
                                            "algorithmic_gbs": round(value * 3 * BYTES_FWD_PER_PATCH * vox * byt / 1e9, 1),
                                            "hbm_frac_of_8TBs": round(value * 3 * BYTES_FWD_PER_PATCH * vox * byt / 1e9 / world / PEAK_HBM_GBS, 4)})(
                (args.size / 64.0) ** 3, 1.0 if args.dtype == "f32" else 1.0 - 0.94 / 2),
        }
        if not args.no_roofline:
            out["roofline"] = conv_kernel_roofline(device, args.batch, args.size, bf16=args.dtype == "bf16")
            # the same kernel family one level down: C = 8 on the half-size volume (up-cell 3, 21 % of the net's FLOPs).  At 64^3 patches
            # that is (2,8,32^3): 1024 one-wave tiles on 1024 SIMDs -- its no-fetch bound is 0.26 (profiles/r05_conv_ab.log)
            out["roofline_c8"] = conv_kernel_roofline(device, args.batch, args.size // 2, bf16=args.dtype == "bf16", c=8)
        if world == 1 and not args.no_kernel_table:
            # what actually dominates the step: top entry points by measured microseconds per step, each against its roofline
            # (recorded on the single-stream schedule: one entry point = one launch of the dependent chain; the replay reuses the
            # recorded pointers, so a failure here must not cost the headline line)
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            try:
                import kernel_table
                side, trainer.side = getattr(trainer, "side", None), None
                try:
                    rows, ncalls = kernel_table.table(lambda: trainer._eager(x, t), device)
                finally:
                    trainer.side = side
                out["roofline_by_time"] = rows
                out["launches_per_step"] = ncalls
                out["roofline_by_time_schedule"] = "single stream, eager (one entry point = one launch of the dependent chain); the headline ran: " + out["config"]["schedule"]
            except Exception as e:
                out["roofline_by_time"] = "failed: %s" % (str(e)[:200],)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.batch, args.size)
        if world == 1 and not args.no_other_configs and args.size == 64 and args.dtype == "f32":
            del trainer
            torch.cuda.empty_cache()
            out["other_configs"] = other_configs(device, args.batch, args.no_graph, args.targets)
    else:
        out = None
    if dist.is_initialized():
        if cm is not None:
            cm.barrier()
        n3d_comm.close_all()
        dist.destroy_process_group()
    if out is not None:
        # RCCL prints its version banner through C stdio, which a pipe only sees at exit: flush it now so that the JSON line is the LAST line
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
