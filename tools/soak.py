#!/usr/bin/env python3
"""Soak of the replayed three-stream schedules: many steps of the benchmark train step and of the search step with fresh inputs every
step; afterwards no hand-off may have timed out, every loss must be finite and the loss must have gone down.
    python tools/soak.py [train steps] [search steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
from nas_3d_unet_amd import searched, nas
from nas_3d_unet_amd.train import Trainer, SearchTrainer
dev = torch.device("cuda")
if os.environ.get("N3D_FORCE_DP") == "1":      # the data-parallel code path on a 1-rank RCCL group (all-reduces through n3d_comm_*)
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29591")
    dist.init_process_group("nccl", rank=0, world_size=1)
n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 5000
n_search = int(sys.argv[2]) if len(sys.argv) > 2 else 300
torch.manual_seed(1234)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=True)
batches = []
for s in range(8):
    xn, tn = bench.synthetic_batch(2, 64, 100 + s)
    batches.append((bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)))
losses = []
for i in range(n_train):
    if i == 1:       # (the first step warms up, captures both schedules and times them)
        torch.cuda.synchronize(); t0 = time.perf_counter()
    x, t = batches[i % 8]
    l = tr.step(x, t)
    if i % 50 == 0 or i == n_train - 1:
        losses.append(float(l))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
tr.check_sync()
assert all(np.isfinite(losses)), losses
print("train: %d replayed steps (8 rotating batches) in %.2f s = %.3f ms per step incl. host; schedule %s; time-outs %d; loss %.4f -> %.4f (min %.4f)"
      % (n_train, dt, dt / max(1, n_train - 1) * 1e3, "side streams" if tr._use_side else "single stream", tr.sync_timeouts(), losses[0], losses[-1], min(losses)))
assert losses[-1] < losses[0]
torch.manual_seed(1234)
snet = nas.ShellNet(4, 4, 3, 4, 3, False, True).to(dev); snet.train()
st = SearchTrainer(snet, graph=True)
vb = []
for s in range(4):
    xn, tn = bench.synthetic_batch(2, 64, 200 + s)
    vb.append((bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)))
la, lw = [], []
for i in range(n_search):
    if i == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    (x, t), (vx, vt) = batches[i % 8], vb[i % 4]
    a, w = st.step(x, t, vx, vt)
    if i % 20 == 0 or i == n_search - 1:
        la.append(float(a)); lw.append(float(w))
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st.check_sync()
assert all(np.isfinite(la)) and all(np.isfinite(lw))
print("search: %d replayed steps in %.2f s = %.2f ms per step incl. host; schedule %s; time-outs %d; weight-pass loss %.4f -> %.4f, architecture-pass loss %.4f -> %.4f"
      % (n_search, dt, dt / max(1, n_search - 1) * 1e3, "side streams" if st._use_side else "single stream", st.sync_timeouts(), lw[0], lw[-1], la[0], la[-1]))
assert lw[-1] < lw[0]
print("soak OK")
if os.environ.get("N3D_FORCE_DP") == "1":
    from nas_3d_unet_amd import comm
    comm.close_all()
    dist.destroy_process_group()
