import sys, os
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R)
import torch
from nas_3d_unet_amd import kernels as K, _lib
dev = torch.device("cuda")
size = 128
x = K.as_view(K.empty_ndhwc(2, 4, size, size, size, dev, torch.float32)); x.t.normal_()
with K.storage(torch.bfloat16):
    dy = K.as_view(K.empty_ndhwc(2, 12, size, size, size, dev)); dy.t.normal_()
w = torch.empty(12, 4, 3, 3, 3, device=dev); dw = torch.empty_like(w)
g = K.conv_geom(2, size, size, size, 4, 12, 3, 1, 1, 1)
for _ in range(5):
    K.conv_bwd_weight(g, x, dy, dw, None, 0, None, False)
torch.cuda.synchronize()
