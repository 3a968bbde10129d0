import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
dev = torch.device("cuda")
def timeit(fn, reps=10, rounds=3):
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        fn(); torch.cuda.synchronize()
        g.capture_begin(capture_error_mode="thread_local")
        for _ in range(reps): fn()
        g.capture_end()
    g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(rounds): g.replay()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) * 1e3 / (reps * rounds)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
for dt in (torch.float32, torch.bfloat16):
    with K.storage(dt):
        for (ci, co, stride) in [(12, 8, 2), (12, 4, 1), (4, 12, 1), (24, 4, 1)]:
            so = S // stride
            x = K.as_view(K.empty_ndhwc(2, ci, S, S, S, dev).normal_())
            y = K.as_view(K.empty_ndhwc(2, co, so, so, so, dev).normal_())
            w = torch.randn(co, ci, 1, 1, 1, device=dev)
            g = K.conv_geom(2, S, S, S, ci, co, 1, stride, 1, 0)
            ctx = K.StepContext(dev)
            with K.step_context(ctx):
                K.conv_fwd(g, x, w, None, y, 0, None, None, False); K.conv_bwd_data(g, y, w, x, 0, None, None, False)
                ctx.freeze(); ctx.pack_all()
                tf = timeit(lambda: K.conv_fwd(g, x, w, None, y, 0, None, None, False))
                td = timeit(lambda: K.conv_bwd_data(g, y, w, x, 0, None, None, False))
            esz = 2 if dt == torch.bfloat16 else 4
            byts = 2 * (S ** 3 * ci + so ** 3 * co) * esz
            print("%s k1 %d->%d s%d %d^3: fwd %.1f us (%.2f TB/s)  dgrad %.1f us (%.2f TB/s)" % ("bf16" if esz == 2 else "fp32", ci, co, stride, S, tf, byts / tf / 1e6, td, byts / td / 1e6))
