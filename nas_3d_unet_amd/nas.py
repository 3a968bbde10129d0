"""Supernet (build-side counterpart of the reference's nas.py:13-135): KernelNet + ShellNet."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .cell import Cell
from .genotype import GenoParser, Genotype
from .prim_ops import ConvOps, DownOps, NormOps, UpOps


class KernelNet(nn.Module):
    """U-shaped stack of supernet cells; module names match the reference for state-dict parity."""

    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change):
        super().__init__()
        assert depth >= 2, "depth must >= 2"
        c0 = c1 = n_nodes * init_n_kernels
        c_node = init_n_kernels
        self.stem0 = ConvOps(in_channels, c0, kernel_size=1, ops_order="weight_norm")
        self.stem1 = ConvOps(in_channels, c1, kernel_size=3, stride=2, ops_order="weight_norm")
        self.down_cells = nn.ModuleList()
        self.up_cells = nn.ModuleList()
        skips = [c0, c1]
        for _ in range(depth):
            c_node = 2 * c_node if channel_change else c_node
            cell = Cell(n_nodes, c0, c1, c_node)
            self.down_cells.append(cell)
            c0, c1 = c1, cell.out_channels
            skips.append(c1)
        skips.pop()
        for _ in range(depth + 1):
            c0 = skips.pop()
            cell = Cell(n_nodes, c0, c1, c_node, downward=False)
            self.up_cells.append(cell)
            c1 = cell.out_channels
            c_node = c_node // 2 if channel_change else c_node
        self.last_conv = nn.Sequential(ConvOps(c1, out_channels, kernel_size=1, dropout_rate=0.1, ops_order="weight"),
                                       nn.Sigmoid())

    def forward(self, x, alpha1_down, alpha1_up, alpha2_down, alpha2_up):
        s0, s1 = self.stem0(x), self.stem1(x)
        stack = [s0, s1]
        for cell in self.down_cells:
            s0, s1 = s1, cell(s0, s1, alpha1_down, alpha2_down)
            stack.append(s1)
        stack.pop()
        for cell in self.up_cells:
            s0 = stack.pop()
            s1 = cell(s0, s1, alpha1_up, alpha2_up)
        return self.last_conv(s1)


class ShellNet(nn.Module):
    """Architecture parameters (four alpha matrices, zero-init) around a KernelNet (nas.py:81-135)."""

    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, normal_w_share=False,
                 channel_change=False):
        super().__init__()
        self.normal_w_share = normal_w_share
        self.n_nodes = n_nodes
        self.kernel = KernelNet(in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change)
        n_edges = sum(range(2, 2 + n_nodes))
        self.alpha2_down = nn.Parameter(torch.zeros((n_edges, len(DownOps))))
        self.alpha2_up = nn.Parameter(torch.zeros((n_edges, len(UpOps))))
        self.alpha1_down = nn.Parameter(torch.zeros((n_edges, len(NormOps))))
        self.alpha1_up = self.alpha1_down if normal_w_share else nn.Parameter(torch.zeros((n_edges, len(NormOps))))
        self._alphas = [(n, p) for n, p in self.named_parameters() if "alpha" in n]

    def alphas(self):
        for _, p in self._alphas:
            yield p

    def forward(self, x):
        sm = lambda a: F.softmax(a, dim=-1)
        return self.kernel(x, sm(self.alpha1_down), sm(self.alpha1_up), sm(self.alpha2_down), sm(self.alpha2_up))

    def get_gene(self):
        parser = GenoParser(self.n_nodes)
        host = lambda a: F.softmax(a, dim=-1).detach().cpu().numpy()
        return Genotype(down=parser.parse(host(self.alpha1_down), host(self.alpha2_down), True),
                        up=parser.parse(host(self.alpha1_up), host(self.alpha2_up), False))
