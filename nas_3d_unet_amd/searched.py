"""Searched network (build-side counterpart of the reference's searched.py:10-111)."""
import torch.nn as nn

from . import fused
from .genotype import Genotype  # noqa: F401  (re-exported like the reference module does)
from .prim_ops import OPS, ConvOps


class SearchedCell(nn.Module):
    """Every node sums exactly two genotype-selected primitives (searched.py:29,45-51)."""

    def __init__(self, n_nodes, c0, c1, c_node, gene, downward=True):
        super().__init__()
        self.n_nodes = n_nodes
        self.c_node = c_node
        self.genolist = gene.down if downward else gene.up
        self.preprocess0 = ConvOps(c0, c_node, kernel_size=1, stride=2 if downward else 1, ops_order="act_weight_norm")
        self.preprocess1 = ConvOps(c1, c_node, kernel_size=1, ops_order="act_weight_norm")
        self._ops = nn.ModuleList([OPS[name](c_node) for name, _ in self.genolist])

    @property
    def out_channels(self):
        return self.n_nodes * self.c_node

    def forward(self, x0, x1):
        """node k = op[2k](xs[i]) + op[2k+1](xs[j]); output = concat of the nodes -- as one fused launch
        program: the second op's epilogue accumulates into the first one's slice of the output buffer."""
        if getattr(self, "_plan", None) is None:
            self._plan = fused.searched_plan(self)
        return fused.SearchedCellFn.apply(self._plan, x0, x1, *self._plan.params)


class SearchedNet(nn.Module):
    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change, gene):
        super().__init__()
        c0 = c1 = n_nodes * init_n_kernels
        c_node = init_n_kernels
        self.stem0 = ConvOps(in_channels, c0, kernel_size=1, ops_order="weight_norm")
        self.stem1 = ConvOps(in_channels, c1, kernel_size=3, stride=2, ops_order="weight_norm")
        self.down_cells = nn.ModuleList()
        self.up_cells = nn.ModuleList()
        skips = [c0, c1]
        for _ in range(depth):
            c_node = 2 * c_node if channel_change else c_node
            cell = SearchedCell(n_nodes, c0, c1, c_node, gene)
            self.down_cells.append(cell)
            c0, c1 = c1, cell.out_channels
            skips.append(c1)
        skips.pop()
        for _ in range(depth + 1):
            c0 = skips.pop()
            cell = SearchedCell(n_nodes, c0, c1, c_node, gene, downward=False)
            self.up_cells.append(cell)
            c1 = cell.out_channels
            c_node = c_node // 2 if channel_change else c_node
        # head dropout is 0.5 for the searched net, 0.1 for the supernet (searched.py:91-93, nas.py:50-52)
        self.last_conv = nn.Sequential(ConvOps(c1, out_channels, kernel_size=1, dropout_rate=0.5, ops_order="weight"),
                                       nn.Sigmoid())

    def forward(self, x):
        s0, s1 = self.stem0(x), self.stem1(x)
        stack = [s0, s1]
        for cell in self.down_cells:
            s0, s1 = s1, cell(s0, s1)
            stack.append(s1)
        stack.pop()
        for cell in self.up_cells:
            s0 = stack.pop()
            s1 = cell(s0, s1)
        return self.last_conv(s1)
