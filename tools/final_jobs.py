import sys, os, collections
sys.path.insert(0, "/root/repo")
import numpy as np, torch
from nas_3d_unet_amd import kernels as K, searched, train, _lib
from oracle import ref_path as orc
size = int(sys.argv[1]) if len(sys.argv) > 1 else 64
storage = sys.argv[2] if len(sys.argv) > 2 else None
cfg = orc.DEFAULT_CFG
net = searched.SearchedNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, cfg.channel_change, searched.Genotype(*orc.G_CONV)).cuda()
tr = train.Trainer(net, graph=False, side_wgrad=False, storage=storage)
rng = np.random.default_rng(1)
x = torch.from_numpy(rng.standard_normal((2, 4, size, size, size)).astype(np.float32)).cuda()
t = torch.from_numpy((rng.uniform(0, 1, (2, 3, size, size, size)) < 0.3).astype(np.float32)).cuda()
lib = _lib.load()
orig = lib.n3d_wgrad_finalize_batch
rows = []
def spy(arr, n, stream):
    for i in range(n):
        j = arr[i]
        if j.nchunks > 0:
            el = j.ntiles * j.ci_t * j.co_t + j.tco * j.co_t
            rows.append((j.nchunks * el * 4, j.nchunks, j.ntiles, j.ci_t, j.co_t, j.Ci, j.Co, j.taps))
    return orig(arr, n, stream)
lib.n3d_wgrad_finalize_batch = spy
tr.step(x, t); rows.clear(); tr.step(x, t)
torch.cuda.synchronize()
tot = sum(r[0] for r in rows)
print("size", size, storage, "jobs", len(rows), "slab bytes read by the finalize launch: %.1f MB" % (tot / 1e6))
agg = collections.Counter()
for r in rows: agg[r[1:]] += r[0]
for k, v in agg.most_common(14): print("  %6.2f MB  nchunks %d ntiles %d tile %dx%d  Ci %d Co %d taps %d" % ((v / 1e6,) + k))
