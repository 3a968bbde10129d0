"""Adam launch alone (1.81 M parameters, the benchmarked net's): usage adam_ab.py"""
import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K
dev = torch.device("cuda")
n = 1813408
p, g, m, v = (torch.randn(n, device=dev) for _ in range(4)); v.abs_()
step = K.step_counter(dev)
def fn():
    K.adam_step(p, g, m, v, step, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1.0)
for _ in range(3): fn()
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph(); st = torch.cuda.Stream()
with torch.cuda.stream(st):
    gr.capture_begin(capture_error_mode="thread_local")
    for _ in range(20): fn()
    gr.capture_end()
gr.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(5): gr.replay()
e1.record(); e1.synchronize()
print("adam_step, %d parameters: %.2f us per launch (%.0f GB/s)" % (n, e0.elapsed_time(e1) * 10, n * 28 / (e0.elapsed_time(e1) * 10) / 1e3))
