"""Phase stamps of the gemm16 kernels (debug build -DG16_STAMP, N3D_LIB=.../libn3d_g16.so): where the time of a deep-level conv goes.
Runs a chain of dependent convs (as the step does) and prints, per phase, the median over workgroups of (stamp - kernel's first stamp)."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from nas_3d_unet_amd import kernels as K, _lib
dev = torch.device("cuda")
lib = _lib.load()
names = ["start", "set-up done", "loads issued", "loads landed", "MFMAs issued", "K-slices reduced", "output stored", "end"]
for (c, size, stride) in [(32, 8, 1), (64, 4, 1), (16, 16, 1), (32, 8, 2), (64, 2, 1)]:
    g = K.conv_geom(2, size, size, size, c, c, 3, stride, 1, 1)
    xs = [K.as_view(K.empty_ndhwc(2, c, size, size, size, dev)) for _ in range(2)]
    for x in xs: x.t.normal_()
    so = (size - 1) // stride + 1
    w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.05
    yv = xs[1] if stride == 1 else K.as_view(K.empty_ndhwc(2, c, so, so, so, dev))
    rows = max(1, int(K.conv_stats_rows(g, False, 0, xs[0], yv)))
    stats = torch.empty((2, rows, c, 2), dtype=torch.float64, device=dev)
    if stride == 1:
        # ping-pong chain: each conv reads what the previous one wrote
        def step(i):
            K.conv_fwd(g, xs[i & 1], w, None, xs[(i + 1) & 1], 0, None, stats, False)
    else:
        y = yv
        def step(i):
            K.conv_fwd(g, xs[0], w, None, y, 0, None, stats, False)
    for i in range(4): step(i)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr.capture_begin(capture_error_mode="thread_local")
        for i in range(10): step(i)
        gr.capture_end()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); e1.synchronize()
    buf = (C.c_ulonglong * (4096 * 16))()
    lib.n3d_debug_g16_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.n3d_debug_g16_stamps(buf, 4096 * 16)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 2, 8).astype(np.int64)
    start = a[:, 0, 0]
    last = (start > 0) & (start >= start.max() - 3000)     # the workgroups of the LAST launch (stamps of earlier launches stay in the buffer)
    a = a[last]
    t0 = a[:, 0, 0].min()
    print("%d->%d %d^3 s%d: %.2f us per conv in the chain (stamped build); %d workgroups; ticks of s_memtime after the kernel's first stamp (median over workgroups):"
          % (c, c, size, stride, e0.elapsed_time(e1) * 100, len(a)))
    for wv, lab in ((0, "wave 0 "), (1, "wave 15")):
        rel = (a[:, wv, :] - t0).astype(np.float64)
        print("   %s " % lab + "  ".join("%s %.0f" % (names[k], float(np.median(rel[:, k]))) for k in range(8)))
    print("   last workgroup ends at %d ticks; workgroup starts spread over %d ticks" % (a[:, :, 7].max() - t0, a[:, 0, 0].max() - t0))
