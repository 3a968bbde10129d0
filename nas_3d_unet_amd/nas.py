"""Supernet: build-side counterpart of the reference's nas.py (`KernelNet` :13-78, `ShellNet` :81-135), same class names,
constructor arguments, parameter names and alpha matrix shapes."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import unet
from .cell import Cell
from .genotype import GenoParser, Genotype
from .prim_ops import DownOps, NormOps, UpOps


class KernelNet(nn.Module):
    """The weights of the supernet: stems, down / up cells of MixedOps, head."""

    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change):
        super().__init__()
        assert depth >= 2, "depth must >= 2"
        specs, head_in = unet.cell_specs(init_n_kernels, depth, n_nodes, channel_change)
        head = unet.build_stems_and_head(self, in_channels, init_n_kernels, out_channels, n_nodes, head_in, head_dropout=0.1)
        cells = [Cell(n_nodes, a, b, width, downward=down) for a, b, width, down in specs]
        self.down_cells = nn.ModuleList(cells[:depth])
        self.up_cells = nn.ModuleList(cells[depth:])
        self.last_conv = head
        # channel counts that are not multiples of 4 (the reference takes any init_n_kernels): the net runs as its zero-padded twin
        self._n3d_padded = unet.needs_padding(init_n_kernels, depth, n_nodes, channel_change)
        self._n3d_ctor = (in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change, None)

    def _n3d_make_twin(self):
        return unet.PaddedTwin(self, "supernet", *self._n3d_ctor, head_dropout=0.1)

    def forward(self, x, alpha1_down, alpha1_up, alpha2_down, alpha2_up):
        # the down cells share (alpha1_down, alpha2_down), the up cells (alpha1_up, alpha2_up)
        if self._n3d_padded:
            return unet.run_padded(self, x, (alpha1_down, alpha1_up, alpha2_down, alpha2_up))
        return unet.run(self, x, (alpha1_down, alpha1_up, alpha2_down, alpha2_up))

    def forward_loss(self, x, t, alpha1_down, alpha1_up, alpha2_down, alpha2_up, smooth=1e-6):
        if self._n3d_padded:
            return unet.run_padded(self, x, (alpha1_down, alpha1_up, alpha2_down, alpha2_up), t, smooth)
        return unet.run_loss(self, x, t, (alpha1_down, alpha1_up, alpha2_down, alpha2_up), smooth)


class ShellNet(nn.Module):
    """Architecture parameters around a KernelNet: four zero-initialised alpha matrices, one row per edge."""

    def __init__(self, in_channels, init_n_kernels, out_channels, depth, n_nodes, normal_w_share=False,
                 channel_change=False):
        super().__init__()
        self.n_nodes, self.normal_w_share = n_nodes, normal_w_share
        self.kernel = KernelNet(in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change)
        rows = sum(k + 2 for k in range(n_nodes))                     # node k has k + 2 incoming edges
        new = lambda prims: nn.Parameter(torch.zeros(rows, len(prims)))
        self.alpha2_down, self.alpha2_up, self.alpha1_down = new(DownOps), new(UpOps), new(NormOps)
        self.alpha1_up = self.alpha1_down if normal_w_share else new(NormOps)
        self._alphas = [(name, p) for name, p in self.named_parameters() if "alpha" in name]

    def alphas(self):
        return (p for _, p in self._alphas)

    def _soft(self):
        return [F.softmax(a, dim=-1) for a in (self.alpha1_down, self.alpha1_up, self.alpha2_down, self.alpha2_up)]

    def forward(self, x):
        return self.kernel(x, *self._soft())

    def forward_loss(self, x, t, smooth=1e-6):
        """(Dice loss, probabilities) with the loss formed inside the head's launches (search.py:224-226,233-235)"""
        return self.kernel.forward_loss(x, t, *self._soft(), smooth=smooth)

    def get_gene(self):
        a1d, a1u, a2d, a2u = (a.detach().cpu().numpy() for a in self._soft())
        decode = GenoParser(self.n_nodes).parse
        return Genotype(down=decode(a1d, a2d, True), up=decode(a1u, a2u, False))
