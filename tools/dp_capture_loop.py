"""Fresh-process loop around the first bucketed data-parallel graph capture (3 of ~75 GPU-suite runs died
with a silent SIGABRT from a non-Python thread inside it).

    python tools/dp_capture_loop.py --runs 300 --out gpurun_out/dp_loop [--mode suite|lean] [--log-level 1] [--env K=V ...]

Every run is a NEW process (this file with --child) that does what tests/test_gpu_dp.py does up to and including the first
bucketed trainer: 1-rank RCCL group, N3D_FORCE_DP=1, [suite mode: the one-bucket graph trainer and the one-bucket eager trainer
first, with their single-GPU references], then Trainer(graph=True, n_buckets=2, side_wgrad="force") for three steps.  The
child runs under tools/bin/libabort_trace.so (LD_PRELOAD): an abort prints the aborting thread's name and native backtrace.
stderr is NOT captured by anything (pytest's fd capture is what made the round-4 aborts silent).  The parent prints one line
per run and a summary; logs of failed runs are kept under --out."""
import argparse
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(mode):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import socket
    import numpy as np
    import torch
    import torch.distributed as dist
    from nas_3d_unet_amd.train import Trainer, reserve_side_streams
    from test_gpu_nets import build_net
    from _util import dev

    reserve_side_streams(torch.device("cuda", 0))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["N3D_FORCE_DP"] = "1"
    dist.init_process_group("nccl", rank=0, world_size=1)
    rng = np.random.default_rng(41)
    x = dev(rng.standard_normal((2, 4, 32, 32, 32)).astype(np.float32))
    t = dev((rng.uniform(0, 1, (2, 3, 32, 32, 32)) < 0.3).astype(np.float32))

    def one(graph, buckets, comm, with_ref):
        sched = "force" if graph else (True if buckets > 1 else None)
        if with_ref:
            os.environ.pop("N3D_FORCE_DP")
            try:
                net, _ = build_net("searched", "G_CONV", 4)
                ref = Trainer(net, graph=graph, side_wgrad=sched)
                lr_ = [float(ref.step(x, t)) for _ in range(3)]
            finally:
                os.environ["N3D_FORCE_DP"] = "1"
        net, _ = build_net("searched", "G_CONV", 4)
        tr = Trainer(net, graph=graph, n_buckets=buckets, comm=comm, side_wgrad=sched)
        assert tr.dp_path and (comm is None or tr.sync.backend == comm)
        l = [float(tr.step(x, t)) for _ in range(3)]
        tr.check_sync()
        if with_ref:
            assert l == lr_, (l, lr_)
        return tr

    comm = os.environ.get("DP_LOOP_COMM") or None      # None: the trainers' default (n3d_comm_* on the shared communicator)
    if mode == "suite":
        one(True, 1, comm, True)
        one(False, 1, comm, True)
    tr = one(True, 2, comm, mode == "suite")
    assert tr._use_side and tr._side_wsegs == 2
    torch.cuda.synchronize()
    dist.destroy_process_group()
    print("child ok", flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", default=None)
    ap.add_argument("--runs", type=int, default=100)
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "dp_loop"))
    ap.add_argument("--mode", default="suite", choices=["suite", "lean"])
    ap.add_argument("--log-level", default="1")
    ap.add_argument("--env", action="append", default=[])
    ap.add_argument("--budget-s", type=float, default=1e9, help="stop starting new runs after this many seconds")
    a = ap.parse_args()
    if a.child:
        return child(a.child)
    os.makedirs(a.out, exist_ok=True)
    env = dict(os.environ)
    env["AMD_LOG_LEVEL"] = a.log_level
    pre = os.path.join(ROOT, "tools", "bin", "libabort_trace.so")
    if os.path.exists(pre):
        env["LD_PRELOAD"] = pre
    for kv in a.env:
        k, v = kv.split("=", 1)
        env[k] = v
    print("dp_capture_loop: runs=%d mode=%s AMD_LOG_LEVEL=%s env=%s preload=%s" % (a.runs, a.mode, a.log_level, a.env, os.path.exists(pre)), flush=True)
    t00, bad, done = time.time(), [], 0
    for i in range(a.runs):
        if time.time() - t00 > a.budget_s:
            break
        log = os.path.join(a.out, "run_%04d.log" % i)
        env["N3D_ABORT_TRACE_FILE"] = os.path.join(a.out, "abort_%04d.txt" % i)
        t0 = time.time()
        with open(log, "wb") as f:
            try:
                rc = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", a.mode], env=env, stdout=f, stderr=subprocess.STDOUT,
                                    timeout=300).returncode
            except subprocess.TimeoutExpired:
                rc = -999
        done += 1
        dt = time.time() - t0
        if rc != 0:
            bad.append((i, rc))
            print("run %4d: rc %d  %.1f s  -> %s" % (i, rc, dt, log), flush=True)
        else:
            os.remove(log)
            if i % 10 == 0:
                print("run %4d: ok  %.1f s" % (i, dt), flush=True)
    print("dp_capture_loop: %d runs, %d failed %s, %.0f s" % (done, len(bad), bad, time.time() - t00), flush=True)


if __name__ == "__main__":
    main()
