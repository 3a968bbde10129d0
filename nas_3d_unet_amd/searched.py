"""Searched network: build-side counterpart of the reference's searched.py (same class names, constructor arguments and
parameter names; `SearchedCell` :10-51, `SearchedNet` :54-111)."""
import torch.nn as nn

from . import fused, unet
from .genotype import Genotype  # noqa: F401  (importable from here, as from the reference module)
from .prim_ops import OPS, ConvOps


class SearchedCell(nn.Module):
    """A cell whose every node is the sum of exactly two genotype-selected primitives."""

    def __init__(self, n_nodes, c0, c1, c_node, gene, downward=True):
        super().__init__()
        self.n_nodes, self.c_node = n_nodes, c_node
        self.genolist = gene.down if downward else gene.up          # [(primitive name, input index)] * 2 * n_nodes
        self.preprocess0 = ConvOps(c0, c_node, kernel_size=1, stride=2 if downward else 1, ops_order="act_weight_norm")
        self.preprocess1 = ConvOps(c1, c_node, kernel_size=1, ops_order="act_weight_norm")
        self._ops = nn.ModuleList(OPS[prim](c_node) for prim, _ in self.genolist)
        self._plan = None

    @property
    def out_channels(self):
        return self.c_node * self.n_nodes

    def forward(self, x0, x1):
        if self.c_node % 4 != 0:
            # odd channel counts outside a build-side net (which runs as one padded twin): op by op, the reference's algebra (searched.py:36-51)
            import torch
            xs = [self.preprocess0(x0), self.preprocess1(x1)]
            for k in range(self.n_nodes):
                xs.append(sum([self._ops[2 * k + j](xs[self.genolist[2 * k + j][1]]) for j in range(2)]))
            return torch.cat(xs[-self.n_nodes:], dim=1)
        # one launch program per cell (fused.py): node k = op[2k](xs[i]) + op[2k+1](xs[j]) written straight into its channel
        # slice of the output buffer, which IS the concatenation the reference builds with torch.cat
        if not fused.current(self._plan):
            self._plan = fused.searched_plan(self)
        return fused.SearchedCellFn.apply(self._plan, x0, x1, *self._plan.params)


class SearchedNet(nn.Module):
    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change, gene):
        super().__init__()
        specs, head_in = unet.cell_specs(init_n_kernels, depth, n_nodes, channel_change)
        head = unet.build_stems_and_head(self, in_channels, init_n_kernels, out_channels, n_nodes, head_in, head_dropout=0.5)
        cells = [SearchedCell(n_nodes, a, b, width, gene, downward=down) for a, b, width, down in specs]
        self.down_cells = nn.ModuleList(cells[:depth])
        self.up_cells = nn.ModuleList(cells[depth:])
        self.last_conv = head                                        # head dropout 0.5 here, 0.1 in the supernet
        # channel counts that are not multiples of 4 (the reference takes any init_n_kernels): the net runs as its zero-padded twin
        self._n3d_padded = unet.needs_padding(init_n_kernels, depth, n_nodes, channel_change)
        self._n3d_ctor = (in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change, gene)

    def _n3d_make_twin(self):
        return unet.PaddedTwin(self, "searched", *self._n3d_ctor, head_dropout=0.5)

    def forward(self, x):
        if self._n3d_padded:
            return unet.run_padded(self, x)
        return unet.run(self, x)

    def forward_loss(self, x, t, smooth=1e-6):
        """(Dice loss of loss.py:12-14, probabilities) with the loss formed inside the head's launches (train.py:121-124)"""
        if self._n3d_padded:
            return unet.run_padded(self, x, None, t, smooth)
        return unet.run_loss(self, x, t, None, smooth)
