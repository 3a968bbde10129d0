"""The launch sequence of the 64^3 train step's side-stream schedule per stream, from the C-ABI calls recorded while the trainer captures it: how many
of the main chain's launches are flag launches (n3d_sync_*) and how many of those sit next to each other (profiles/r06_handoff_seq.log)."""
import os, sys, collections
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tools")
import torch, bench
from kernel_table import Recorder
from nas_3d_unet_amd import searched
from nas_3d_unet_amd.train import Trainer, reserve_side_streams
dev = torch.device("cuda", 0)
reserve_side_streams(dev)
torch.manual_seed(1234)
net = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(**bench.G_CONV)).to(dev); net.train()
tr = Trainer(net, graph=True)
xn, tn = bench.synthetic_batch(2, 64, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
with Recorder() as r:
    tr.step(x, t)
torch.cuda.synchronize()
calls = r.calls
print("calls recorded:", len(calls))
# passes are separated by n3d_pack_batch (one per forward); take the LAST pass that contains sync calls
starts = [i for i, (n, a) in enumerate(calls) if n == "n3d_pack_batch"] + [len(calls)]
best = None
for a, b in zip(starts[:-1], starts[1:]):
    seg = calls[a:b]
    ns = sum(1 for n, _ in seg if n.startswith("n3d_sync"))
    print("pass at", a, "len", b - a, "sync calls", ns)
    if ns: best = seg
def sp(args):
    v = args[-1]
    return getattr(v, "value", v)
streams = collections.OrderedDict()
for n, a in best:
    streams.setdefault(sp(a), []).append(n)
for s, names in streams.items():
    sync = [n for n in names if n.startswith("n3d_sync")]
    print("stream", s, "launches", len(names), "sync", len(sync), collections.Counter(sync))
main = max(streams.items(), key=lambda kv: len(kv[1]))[1]
# adjacency on the main stream
adj = collections.Counter()
for p, q in zip(main[:-1], main[1:]):
    if p.startswith("n3d_sync") and q.startswith("n3d_sync"):
        adj[(p, q)] += 1
print("adjacent sync pairs on the main stream:", dict(adj))
runs = []; cur = 0
for n in main:
    if n.startswith("n3d_sync"): cur += 1
    else:
        if cur: runs.append(cur)
        cur = 0
print("runs of consecutive sync launches on main:", collections.Counter(runs))
print("main sequence (S=signal W=wait 2=wait2 .=other):")
print("".join("S" if n == "n3d_sync_signal" else "W" if n == "n3d_sync_wait" else "2" if n == "n3d_sync_wait2" else "." for n in main))
