"""MI355X-native drop-in for the reference's ``cell`` module (cell.py:8-82): MixedOp and Cell."""
import torch.nn as nn

from . import fused
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
        if getattr(self, "_segs", None) is None:
            self._segs = [fused._single_segment(op) for op in self._ops]
            self._plist = [p for seg in self._segs for p in seg.params()]
        # one autograd node: every primitive's epilogue accumulates w_k * op_k(x) into the same buffer
        return fused.MixedOpFn.apply(self._segs, x, weights, *self._plist)


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
        """x0, x1: cell inputs; alpha1 / alpha2: full (n_edges, n_prims) weight matrices for the stride-1 /
        stride-2 edges, both indexed by the global edge counter (cell.py:76-80).  Runs as one fused launch
        program (fused.py); the result is the channel concat of the n_nodes node outputs."""
        if getattr(self, "_plan", None) is None:
            self._plan = fused.supernet_plan(self)
        return fused.CellFn.apply(self._plan, x0, x1, alpha1, alpha2, *self._plan.params)
