"""Time n3d_pack_batch / n3d_wgrad_finalize_batch of a real train step inside a HIP graph, for several builds of the library."""
import sys, os, ctypes as C, glob
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
import bench
from nas_3d_unet_amd import searched, kernels as K, _lib
from nas_3d_unet_amd.train import Trainer
dev = torch.device("cuda:0")
Cf = bench.CFG
net = searched.SearchedNet(Cf["in_channels"], Cf["init_n_kernels"], Cf["out_channels"], Cf["depth"], Cf["n_nodes"], Cf["channel_change"],
                           searched.Genotype(**bench.G_CONV)).to(dev)
net.train()
tr = Trainer(net, graph=False)
xn, tn = bench.synthetic_batch(2, 64, 1)
x, t = bench.to_patch_layout(torch.from_numpy(xn).to(dev)), torch.from_numpy(tn).to(dev)
tr.step(x, t); tr.step(x, t)
torch.cuda.synchronize()
ctx = tr.ctx
print("pack jobs", ctx.njobs)

def timeit(fn, reps=20, rounds=5):
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        sp = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn(sp); torch.cuda.synchronize()
        g.capture_begin(capture_error_mode="thread_local")
        for _ in range(reps): fn(sp)
        g.capture_end()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)

# the deferred reductions of one step: record them instead of running them
rec = []
orig = K.StepContext.flush_final
def grab(self):
    self.join()
    rec.extend(self.final)
    return orig(self)
K.StepContext.flush_final = grab
tr.step(x, t)
torch.cuda.synchronize()
K.StepContext.flush_final = orig
arr = (_lib.FinalJob * len(rec))(*rec)
print("final jobs", len(rec), "partial floats", sum(j.nchunks * j.ntiles * j.ci_t * j.co_t for j in rec))
for path in sorted(glob.glob(os.path.join(R, "tools/bin/libn3d_[PF]*.so"))):
    lib = _lib.load(path)
    us = timeit(lambda sp: _lib.check(lib.n3d_pack_batch(ctx.jobs, ctx.njobs, sp)))
    uf = timeit(lambda sp: _lib.check(lib.n3d_wgrad_finalize_batch(arr, len(rec), sp)))
    print("%-40s pack_batch %.2f us   finalize_batch %.2f us" % (os.path.basename(path), us, uf))
import collections
groups = collections.OrderedDict()
for j in rec:
    key = "direct<=4" if j.nchunks <= 4 else ("direct<=16" if j.nchunks <= 16 else ("many<=64" if j.nchunks <= 64 else ("many<=128" if j.nchunks <= 128 else "many>128")))
    groups.setdefault(key, []).append(j)
for path in sorted(glob.glob(os.path.join(R, "tools/bin/libn3d_F[012].so"))):
    lib = _lib.load(path)
    for key, js in groups.items():
        a2 = (_lib.FinalJob * len(js))(*js)
        uf = timeit(lambda sp: _lib.check(lib.n3d_wgrad_finalize_batch(a2, len(js), sp)))
        fl = sum(j.nchunks * j.ntiles * j.ci_t * j.co_t for j in js)
        print("%-16s %-12s jobs %3d partial floats %8d  %.2f us" % (os.path.basename(path), key, len(js), fl, uf))
