"""Round 6: HOST time of one replayed train step under the data-parallel code path, call by call.
A 1-rank RCCL group (N3D_FORCE_DP=1) runs exactly what N > 1 runs: with `--buckets 2` a step is N raw graph launches (the side
stream's head + weight-gradient segments, the main chain, the tail), the bucket all-reduces between them, the guard flag, the last
all-reduce and the guarded Adam.  Eight processes each pay this on their own core; the budget is the 1.6 ms step.
   python tools/dp_host_budget.py [buckets] [steps]"""
import os, sys, time, collections
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ.setdefault("N3D_FORCE_DP", "1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from nas_3d_unet_amd import kernels as K, searched, train
from oracle import ref_path as orc      # (genotype / config constants only)

buckets = int(sys.argv[1]) if len(sys.argv) > 1 else 2
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 300
dist.init_process_group("nccl", rank=0, world_size=1)
torch.manual_seed(0)
cfg = orc.DEFAULT_CFG
net = searched.SearchedNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, cfg.channel_change,
                           searched.Genotype(*orc.G_CONV)).cuda()
tr = train.Trainer(net, graph=True, n_buckets=buckets, side_wgrad="force")
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.standard_normal((2, 4, 64, 64, 64)).astype(np.float32)).cuda()
t = torch.from_numpy((rng.uniform(0, 1, (2, 3, 64, 64, 64)) < 0.3).astype(np.float32)).cuda()
for _ in range(10):
    tr.step(x, t)
torch.cuda.synchronize()
x, t = tr.input_buffers()          # no per-step input copy (a data step writes the batch into the trainer's own buffers)
acc, cnt = collections.defaultdict(float), collections.Counter()


def timed(obj, name, label):
    fn = getattr(obj, name)

    def wrapper(*a, **k):
        t0 = time.perf_counter()
        r = fn(*a, **k)
        acc[label] += time.perf_counter() - t0
        cnt[label] += 1
        return r
    setattr(obj, name, wrapper)


timed(K, "graph_launch", "raw graph launch (side stream head / weight-gradient segments)")
timed(tr.sync, "reduce_range", "bucket all-reduce (ncclAllReduce on a stream)")
timed(K, "guard_flag", "guard flag launch")
timed(tr.fp, "adam", "guarded Adam launch")
g_main, side_exec, g_tail = tr._side_graphs
for g, lab in ((g_main, "main-chain graph launch"), (g_tail, "tail graph launch")):
    timed(g, "replay", lab)
per_step = []
for i in range(steps):
    if i % 50 == 0:
        torch.cuda.synchronize()       # never measure a full launch queue
    t0 = time.perf_counter()
    tr.step(x, t)
    per_step.append(time.perf_counter() - t0)
torch.cuda.synchronize()
tr.check_sync()
tot = np.array(per_step) * 1e6
print("data-parallel code path on a 1-rank RCCL group, %d bucket(s), side-stream schedule, %d replayed steps" % (len(tr.sync.ranges), steps))
print("host time of Trainer.step(): median %.1f us, mean %.1f us, p99 %.1f us  (the step itself: ~1630 us of GPU time)" % (np.median(tot), tot.mean(), np.percentile(tot, 99)))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-70s %5.1f calls/step  %6.1f us/step  (%.1f us each)" % (k, cnt[k] / steps, v / steps * 1e6, v / cnt[k] * 1e6))
print("  %-70s %18s %6.1f us/step" % ("everything else in step() (polls, replay monitor, Python)", "", tot.mean() - sum(acc.values()) / steps * 1e6))
