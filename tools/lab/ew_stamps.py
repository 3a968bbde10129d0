"""Phase stamps of n3d_affine_act_bwd_apply_gn2 (debug build -DEW_STAMP, N3D_LIB=.../libn3d_ew.so) in a reduce2 -> apply_gn2 chain."""
import sys, os, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import numpy as np, torch
from nas_3d_unet_amd import kernels as K, _lib
dev = torch.device("cuda")
lib = _lib.load()
names = ["start", "first loads out", "prologue 0", "prologue 1", "rest of loads out", "loads landed", "math + stores out", "stores done"]
if os.environ.get("EW_R2"):   # a -DEW_STAMP -DEW_STAMP_R2 build: the stamps are those of reduce2
    names = ["start", "loads + sums done", "class sums (both terms)", "-", "-", "-", "rows written", "-"]
K.SMALL_NODE_BACKWARD = False
for (c, shape) in [(32, (8, 8, 8)), (16, (16, 16, 16)), (64, (4, 4, 4)), (8, (32, 32, 32))]:
    B = 2
    G = 1 if c % 16 else c // 16
    N = shape[0] * shape[1] * shape[2]
    rv = [K.as_view(torch.randn(B, c, *shape, device=dev).contiguous(memory_format=torch.channels_last_3d)) for _ in range(2)]
    gp = [torch.nn.Parameter(torch.randn(c, device=dev)) for _ in range(2)]
    bp = [torch.nn.Parameter(torch.randn(c, device=dev)) for _ in range(2)]
    sv = []
    for k in range(2):
        st, rows = K.channel_stats(rv[k])
        sv.append(K.gn_coeffs(st, rows, gp[k], bp[k], B, c, G, N, 1e-5))
    dv = K.as_view(torch.randn(B, c, *shape, device=dev).contiguous(memory_format=torch.channels_last_3d))
    draws = [K.as_view(K.empty_ndhwc(B, c, *shape, dev)) for _ in range(2)]
    def step():
        tl = [dict(raw=rv[k], a=sv[k][0], b=sv[k][1], mr=sv[k][2], sumraw=sv[k][3], gamma=gp[k], beta=bp[k], wptr=None, relu=True,
                   conv_bias=None, draw=draws[k]) for k in range(2)]
        K.affine_act_bwd_gn2(dv, tl, G)
    for _ in range(3): step()
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        gr.capture_begin(capture_error_mode="thread_local")
        for _ in range(10): step()
        gr.capture_end()
    gr.replay(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); gr.replay(); e1.record(); e1.synchronize()
    buf = (C.c_ulonglong * (4096 * 8))()
    lib.n3d_debug_ew_stamps.argtypes = [C.c_void_p, C.c_int]
    lib.n3d_debug_ew_stamps(buf, 4096 * 8)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 8).astype(np.int64)
    start = a[:, 0]
    last = (start > 0) & (start >= start.max() - 20000)
    a = a[last]
    t0 = a[:, 0].min()
    rel = (a - t0).astype(np.float64)
    print("(2,%d,%d) reduce2 + apply_gn2: %.2f us per pair (stamped build); apply_gn2: %d workgroups; ticks after the first stamp, median [max] over workgroups:" % (N, c, e0.elapsed_time(e1) * 100, len(a)))
    print("   " + "  ".join("%s %.0f [%.0f]" % (names[k], float(np.median(rel[:, k])), float(rel[:, k].max())) for k in range(8)))
