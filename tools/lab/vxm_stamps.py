"""Phase stamps of the marching bf16 conv (debug build -DVXM_STAMP, N3D_LIB=tools/build/libn3d_VXM_STAMP.so): wall-clock stamps (100 MHz)
of every wave of one launch at (2,4,128^3): start, prologue fill issued / landed, per step: compute done / closing wait done, end."""
import os, sys, ctypes as C
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tools"))
import numpy as np, torch
from nas_3d_unet_amd import kernels as K, _lib
dev = torch.device("cuda", 0)
size, B, c = 128, 2, 4
DT = torch.bfloat16
x = K.as_view(K.empty_ndhwc(B, c, size, size, size, dev, DT).normal_())
y = K.as_view(K.empty_ndhwc(B, c, size, size, size, dev, DT).normal_())
w = torch.randn(c, c, 3, 3, 3, device=dev) * 0.1
b = torch.randn(c, device=dev) * 0.1
g = K.conv_geom(B, size, size, size, c, c, 3, 1, 1, 1)
rows = K.conv_stats_rows(g, False, 0, x, y)
stats = torch.empty((B, rows, c, 2), dtype=torch.float64, device=dev)
for _ in range(5):
    K.conv_fwd(g, x, w, b, y, 0, None, stats, False)
torch.cuda.synchronize()
lib = C.CDLL(_lib.LIB_PATH)
n = 4096 * 16
buf = (C.c_ulonglong * n)()
assert lib.n3d_debug_vxm_stamps(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, 16).astype(np.int64)
live = a[:, 0] > 0
a = a[live]
t0 = a[:, 0].min()
us = lambda v: v / 100.0      # 100 MHz wall clock
print("waves stamped:", len(a))
print("start spread   : first %.2f us, last %.2f us after the first wave's start" % (0.0, us(a[:, 0].max() - t0)))
print("end            : first wave done %.2f us, last %.2f us" % (us(a[:, 13].min() - t0), us(a[:, 13].max() - t0)))
print("prologue issue : %.2f us (mean)   fill wait %.2f us (mean, max %.2f)" % (us((a[:, 1] - a[:, 0]).mean()), us((a[:, 2] - a[:, 1]).mean()), us((a[:, 2] - a[:, 1]).max())))
prev = a[:, 2]
for s in range(4):
    cdone, wdone = a[:, 3 + 2 * s], a[:, 4 + 2 * s]
    ok = cdone > 0
    if not ok.any():
        break
    print("step %d: compute %.2f us (mean; min %.2f max %.2f)   closing wait %.2f us (mean, max %.2f)" %
          (s, us((cdone - prev)[ok].mean()), us((cdone - prev)[ok].min()), us((cdone - prev)[ok].max()), us((wdone - cdone)[ok].mean()), us((wdone - cdone)[ok].max())))
    prev = wdone
print("wave lifetime  : mean %.2f us, min %.2f, max %.2f" % (us((a[:, 13] - a[:, 0]).mean()), us((a[:, 13] - a[:, 0]).min()), us((a[:, 13] - a[:, 0]).max())))
hw = a[:, 15]
cu = (hw >> 8) & 0xf; se = (hw >> 13) & 0x7; simd = (hw >> 4) & 0x3; xcc = (hw >> 32) & 0xf
key = xcc * 100000 + se * 1000 + cu * 10 + simd
u, cnt = np.unique(key, return_counts=True)
print("waves per SIMD : %d SIMDs used, min %d max %d waves" % (len(u), cnt.min(), cnt.max()), np.bincount(cnt))
