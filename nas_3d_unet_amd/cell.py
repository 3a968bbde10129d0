"""MI355X-native drop-in for the reference's ``cell`` module (cell.py:8-82): MixedOp and Cell."""
import torch
import torch.nn as nn

from .prim_ops import OPS, ConvOps, DownOps, NormOps, UpOps


class MixedOp(nn.Module):
    """DARTS continuous relaxation: sum_k w_k * op_k(x) over the edge's primitive list (cell.py:9-33)."""

    def __init__(self, channels, stride, transposed=False):
        super().__init__()
        self._ops = nn.ModuleList()
        self.stride = stride
        names = NormOps if stride == 1 else (UpOps if transposed else DownOps)
        for name in names:
            self._ops.append(OPS[name](channels))

    def forward(self, x, alpha1, alpha2):
        weights = alpha1 if self.stride == 1 else alpha2
        return sum(w * op(x) for w, op in zip(weights, self._ops))


class Cell(nn.Module):
    """Supernet cell: n_nodes nodes, node n has n+2 incoming MixedOp edges (cell.py:36-82)."""

    def __init__(self, n_nodes, c0, c1, c_node, downward=True):
        super().__init__()
        self.n_nodes = n_nodes
        self.c_node = c_node
        self.preprocess0 = ConvOps(c0, c_node, kernel_size=1, stride=2 if downward else 1, ops_order="act_weight_norm")
        self.preprocess1 = ConvOps(c1, c_node, kernel_size=1, ops_order="act_weight_norm")
        self._ops = nn.ModuleList()
        for n_in in range(2, 2 + n_nodes):
            for i in range(n_in):
                if downward:
                    self._ops.append(MixedOp(c_node, stride=2 if i <= 1 else 1))
                else:
                    self._ops.append(MixedOp(c_node, stride=2 if i == 1 else 1, transposed=True))

    @property
    def out_channels(self):
        return self.n_nodes * self.c_node

    def forward(self, x0, x1, alpha1, alpha2):
        xs = [self.preprocess0(x0), self.preprocess1(x1)]
        e = 0
        for _ in range(self.n_nodes):
            acc = 0
            for x in list(xs):
                acc = acc + self._ops[e](x, alpha1[e], alpha2[e])
                e += 1
            xs.append(acc)
        return torch.cat(xs[-self.n_nodes:], dim=1)
