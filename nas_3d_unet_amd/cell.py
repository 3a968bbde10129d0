"""MI355X-native drop-in for the reference's ``cell`` module (`MixedOp` cell.py:8-33, `Cell` :35-82): same class names,
constructor arguments, forward signatures and `_ops` / `preprocess*` attribute names."""
import torch.nn as nn

from . import fused
from .prim_ops import OPS, ConvOps, DownOps, NormOps, UpOps


def edge_kinds(n_nodes, downward):
    """(stride, transposed) of every edge in the cell's edge order: node k has k + 2 inputs (the two cell inputs, then the
    earlier nodes).  Down cell: the edges from the two cell inputs reduce (stride 2); up cell: only the edge from the second
    cell input expands (stride-2 transposed primitives)."""
    kinds = []
    for n_inputs in range(2, n_nodes + 2):
        for src in range(n_inputs):
            strided = src < 2 if downward else src == 1
            kinds.append((2 if strided else 1, not downward))
    return kinds


class MixedOp(nn.Module):
    """DARTS relaxation of one edge: sum_k w_k * op_k(x) over the primitive list the stride selects."""

    def __init__(self, channels, stride, transposed=False):
        super().__init__()
        self.stride = stride
        prims = NormOps if stride == 1 else (UpOps if transposed else DownOps)
        self._ops = nn.ModuleList(OPS[p](channels) for p in prims)
        self._segs = None

    def forward(self, x, alpha1, alpha2):
        if self._ops[0]._n3d_io[0] % 4 != 0:
            # channel counts that are not multiples of 4: every primitive runs through its zero-padded twin (prim_ops._OpTwin) and the
            # weighted sum is the reference's own expression (cell.py:24-32)
            alpha = alpha1 if self.stride == 1 else alpha2
            return sum([w * op(x) for w, op in zip(alpha, self._ops)])
        if self._segs is None:
            self._segs = [fused._single_segment(op) for op in self._ops]
            self._plist = [p for seg in self._segs for p in seg.params()]
        # one autograd node; every primitive's epilogue accumulates w_k * op_k(x) into the same buffer
        return fused.MixedOpFn.apply(self._segs, x, alpha1 if self.stride == 1 else alpha2, *self._plist)


class Cell(nn.Module):
    """Supernet cell.  alpha1 / alpha2 are the full (n_edges, n_prims) matrices of the stride-1 / stride-2 edges, both
    indexed by the cell-wide edge counter (so each matrix has rows that are never read)."""

    def __init__(self, n_nodes, c0, c1, c_node, downward=True):
        super().__init__()
        self.n_nodes, self.c_node = n_nodes, c_node
        self.preprocess0 = ConvOps(c0, c_node, kernel_size=1, stride=2 if downward else 1, ops_order="act_weight_norm")
        self.preprocess1 = ConvOps(c1, c_node, kernel_size=1, ops_order="act_weight_norm")
        self._ops = nn.ModuleList(MixedOp(c_node, stride=s, transposed=t) for s, t in edge_kinds(n_nodes, downward))
        self._plan = None

    @property
    def out_channels(self):
        return self.c_node * self.n_nodes

    def forward(self, x0, x1, alpha1, alpha2):
        if self.c_node % 4 != 0:
            # channel counts that are not multiples of 4 (a caller that builds its own net from these cells: the reference's unchanged
            # nas.py with init_n_kernels = 6, say): op by op through the zero-padded twins, the cell algebra as the reference states it
            # (cell.py:67-82).  The build-side nets run such a net as ONE padded twin instead (unet.PaddedTwin): faster, same results.
            import torch
            xs = [self.preprocess0(x0), self.preprocess1(x1)]
            i = 0
            for _ in range(self.n_nodes):
                outputs = []
                for x in xs:
                    outputs.append(self._ops[i](x, alpha1[i], alpha2[i]))
                    i += 1
                xs.append(sum(outputs))
            return torch.cat(xs[-self.n_nodes:], dim=1)
        # one launch program per cell (fused.py); the result is the channel concatenation of the node outputs
        if not fused.current(self._plan):
            self._plan = fused.supernet_plan(self)
        return fused.CellFn.apply(self._plan, x0, x1, alpha1, alpha2, *self._plan.params)
