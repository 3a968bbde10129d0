"""U-shaped wiring shared by the supernet (nas.KernelNet) and the searched net (searched.SearchedNet).

The reference builds the same encoder / decoder twice (nas.py:25-78 and searched.py:66-111): two stems, `depth` down
cells whose node width doubles when `channel_change`, `depth + 1` up cells fed by a stack of skip tensors, a 1x1x1 head.
Here it is one channel plan and one routing function; the two nets differ only in the cell type and in what a cell call
passes along (alpha matrices or nothing)."""
import torch.nn as nn

from . import fused, head as _head
from .prim_ops import ConvOps


def cell_specs(init_n_kernels, depth, n_nodes, channel_change):
    """([(c0, c1, c_node, downward)] for the down cells then the up cells, channels entering the head)"""
    wide = n_nodes * init_n_kernels          # both stems emit n_nodes * init_n_kernels channels
    specs, pending = [], [wide, wide]        # `pending`: channel counts of the skip tensors, in push order
    prev2, prev1, node_c = wide, wide, init_n_kernels
    for _ in range(depth):
        if channel_change:
            node_c *= 2
        specs.append((prev2, prev1, node_c, True))
        prev2, prev1 = prev1, n_nodes * node_c
        pending.append(prev1)
    pending.pop()                            # the deepest output goes straight into the first up cell
    for _ in range(depth + 1):
        specs.append((pending.pop(), prev1, node_c, False))
        prev1 = n_nodes * node_c
        if channel_change:
            node_c //= 2
    return specs, prev1


def needs_padding(init_n_kernels, depth, n_nodes, channel_change):
    """does any feature map of this net have a channel count that is not a multiple of 4?  The kernels move channels four at a time
    (16-byte NDHWC vectors); such a net runs as its zero-padded twin (PaddedTwin)."""
    specs, _ = cell_specs(init_n_kernels, depth, n_nodes, channel_change)
    return any(c % 4 for c in {n_nodes * init_n_kernels} | {spec[2] for spec in specs})


def _pad4(c):
    return (c + 3) // 4 * 4


class PaddedTwin:
    """A net whose feature-map channel counts are not multiples of 4 (the reference takes any init_n_kernels: nas.py:13-26,
    searched.py:55-66) runs as a TWIN built by the same constructors with every feature map zero-padded to the next multiple of 4:
    stems n c -> pad4(n c), node width c -> pad4(c), a cell output n c -> n pad4(c) (node by node).  The padded channels carry zero conv
    weights, zero GroupNorm gamma / beta and zero SE weights, so they hold exactly 0 in every forward tensor, and nothing of what flows back
    through them reaches a real channel or a real parameter (a padded channel's activation gradient is NOT zero -- GroupNorm couples the
    channels of a group -- but it only ever meets zero weights; the gradients of the padded PARAMETERS are discarded: extract() here, a mask
    in front of Adam in train.Trainer); the real channels see the reference's arithmetic -- with ONE correction: a GroupNorm group's element count is that of the
    REAL channels (programs.gn_groups -> a negative group count at the C ABI, include/n3d.h "padded channels").  The user-visible module
    keeps the reference's parameter shapes (state-dict parity); before every forward they are embedded into the twin's parameters and
    after the backward the twin's gradients are cut back (index plumbing in torch, arithmetic in libn3d).  That is the module API (forward /
    forward_loss under autograd, unet.run_padded); the flat-buffer trainers train the twin itself (padded entries masked in front of Adam)
    and hand the parameters back in the reference's shapes at check_sync() (train.Trainer / SearchTrainer)."""

    def __init__(self, real, kind, in_channels, init_n_kernels, out_channels, depth, n_nodes, channel_change, gene=None, head_dropout=0.0):
        import torch
        from .cell import Cell
        from .searched import SearchedCell
        self.n_nodes = n_nodes
        specs, head_in = cell_specs(init_n_kernels, depth, n_nodes, channel_change)
        wide = n_nodes * init_n_kernels
        # which inputs are cell outputs (node-wise padding) and which are stem outputs (padded at the end): unet.route's wiring
        kinds, older, newer, kept = [], "stem", "stem", ["stem", "stem"]
        for _ in range(depth):
            kinds.append((older, newer))
            older, newer = newer, "cell"
            kept.append("cell")
        kept.pop()
        for _ in range(depth + 1):
            kinds.append((kept.pop(), newer))
            newer = "cell"
        self.kinds = kinds
        padded_in = lambda c, kind_: _pad4(c) if kind_ == "stem" else n_nodes * _pad4(c // n_nodes)
        twin = nn.Module()
        twin.stem0 = ConvOps(in_channels, _pad4(wide), kernel_size=1, ops_order="weight_norm")
        twin.stem1 = ConvOps(in_channels, _pad4(wide), kernel_size=3, stride=2, ops_order="weight_norm")
        cells = []
        for (c0, c1, width, down), (k0, k1) in zip(specs, kinds):
            a, b, w = padded_in(c0, k0), padded_in(c1, k1), _pad4(width)
            cells.append(SearchedCell(n_nodes, a, b, w, gene, downward=down) if kind == "searched" else Cell(n_nodes, a, b, w, downward=down))
        twin.down_cells = nn.ModuleList(cells[:depth])
        twin.up_cells = nn.ModuleList(cells[depth:])
        twin.last_conv = nn.Sequential(ConvOps(n_nodes * _pad4(head_in // n_nodes), out_channels, kernel_size=1, dropout_rate=head_dropout,
                                               ops_order="weight"), nn.Sigmoid())
        dev = next(real.parameters()).device
        self.twin = twin.to(dev)
        for p in self.twin.parameters():
            p.requires_grad_(True)
        # GroupNorms remember the real channel count of their tensor
        rmods = dict(real.named_modules())
        for name, m in self.twin.named_modules():
            if isinstance(m, nn.GroupNorm):
                m._n3d_real_c = int(rmods[name].num_channels)
        # per parameter and dimension: positions of the real entries inside the twin's (None = same size)
        self.names = [n for n, _ in real.named_parameters() if not n.startswith("alpha")]
        tp = dict(self.twin.named_parameters())
        assert sorted(tp) == sorted(self.names), "padded twin: parameter names differ from the net's"
        self.maps = {}
        rp = dict(real.named_parameters())
        for n in self.names:
            r, t = rp[n], tp[n]
            idx = []
            for d, (rs, ts) in enumerate(zip(r.shape, t.shape)):
                if rs == ts:
                    idx.append(None)
                    continue
                block = False
                if d == 1 and n.endswith("conv.weight"):
                    # the input side of a conv that reads a CELL output: node-wise padding
                    if n.startswith("last_conv."):
                        block = True
                    elif ".preprocess" in n:
                        ci = int(n.split(".")[1]) + (0 if n.startswith("down_cells.") else depth)
                        block = kinds[ci][0 if ".preprocess0." in n else 1] == "cell"
                if block:
                    c = rs // n_nodes
                    assert rs == n_nodes * c and ts == n_nodes * _pad4(c), (n, rs, ts)
                    pos = torch.cat([k * _pad4(c) + torch.arange(c) for k in range(n_nodes)])
                else:
                    assert ts == _pad4(rs), (n, d, rs, ts)
                    pos = torch.arange(rs)
                idx.append(pos.to(dev))
            self.maps[n] = idx

    def embed(self, real):
        """real parameters -> the twin's (zeros elsewhere); module state that is not a parameter follows too"""
        import torch
        tp = dict(self.twin.named_parameters())
        with torch.no_grad():
            for n, r in real.named_parameters():
                if n in self.maps:
                    tp[n].copy_(self.embed_tensor(n, r.detach(), tp[n].shape))
        self.twin.train(real.training)
        rmods = dict(real.named_modules())
        for name, m in self.twin.named_modules():
            rm = rmods.get(name)
            if rm is not None and hasattr(rm, "dropout") and hasattr(m, "dropout") and (rm.dropout is None) != (m.dropout is None):
                m.dropout = None if rm.dropout is None else nn.Dropout3d(rm.dropout.p)

    def embed_tensor(self, name, r, shape):
        """a tensor of the real parameter `name`'s shape -> the twin parameter's `shape`, zeros at the padded positions"""
        import torch
        cur = r
        for d, pos in enumerate(self.maps[name]):
            if pos is None:
                continue
            sh = list(cur.shape)
            sh[d] = shape[d]
            new = torch.zeros(sh, dtype=cur.dtype, device=cur.device)
            new.index_copy_(d, pos, cur)
            cur = new
        return cur

    def extract(self, name, g):
        """a gradient of the twin's parameter `name` -> the real parameter's shape"""
        for d, pos in enumerate(self.maps[name]):
            if pos is not None:
                g = g.index_select(d, pos)
        return g


class _padded_switches:
    """while kernels of a padded twin are launched: conv-bias gradients by summation, per-term GroupNorm launches (the node-level ones
    share their element count with SE gates), no node-planar inner cells (train._padded_flags is the same set)"""

    def __enter__(self):
        from . import programs as P
        self.prev = (P.ANALYTIC_CONV_BIAS, fused.NODE_PHASES, fused.NODE_APPLY, P.NODE_FWD_COEFFS, fused.PLANAR_INNER)
        P.ANALYTIC_CONV_BIAS, fused.NODE_PHASES, fused.NODE_APPLY, P.NODE_FWD_COEFFS, fused.PLANAR_INNER = False, False, False, False, False

    def __exit__(self, *exc):
        from . import programs as P
        P.ANALYTIC_CONV_BIAS, fused.NODE_PHASES, fused.NODE_APPLY, P.NODE_FWD_COEFFS, fused.PLANAR_INNER = self.prev
        return False


def _padded_net_fn():
    """the autograd node of run_padded (built on first use: torch is imported lazily in this module)"""
    global _PaddedNetFn
    if _PaddedNetFn is not None:
        return _PaddedNetFn
    import torch

    class PaddedNetFn(torch.autograd.Function):
        """inputs: (net, twin, need_grad, loss target | None, smooth, number of alphas, x, *alphas, *the net's parameters in twin.names order)"""

        @staticmethod
        def forward(ctx, net, tw, need_grad, loss_target, smooth, n_al, xin, *rest):
            al = rest[:n_al]
            tw.embed(net)
            with _padded_switches(), torch.set_grad_enabled(need_grad):
                xi = xin.detach().requires_grad_(need_grad and xin.requires_grad)
                ali = tuple(a.detach().requires_grad_(need_grad and a.requires_grad) for a in al)
                if loss_target is None:
                    out = (run(tw.twin, xi, ali if n_al else None),)
                else:
                    out = run_loss(tw.twin, xi, loss_target, ali if n_al else None, smooth)
            if need_grad:
                ctx.saved = (tw, xi, ali, out)
            return tuple(o.detach() for o in out)

        @staticmethod
        def backward(ctx, *douts):
            tw, xi, ali, out = ctx.saved
            tp = dict(tw.twin.named_parameters())
            wanted = [xi] if xi.requires_grad else []
            wanted += [a for a in ali if a.requires_grad]
            wanted += [tp[n] for n in tw.names]
            outs = [o for o, d in zip(out, douts) if d is not None and o.requires_grad]
            gouts = [d for o, d in zip(out, douts) if d is not None and o.requires_grad]
            with _padded_switches():
                gs = list(torch.autograd.grad(outs, wanted, gouts, allow_unused=True))
            gx = gs.pop(0) if xi.requires_grad else None
            gal = [gs.pop(0) if a.requires_grad else None for a in ali]
            gpar = [tw.extract(n, g) if g is not None else None for n, g in zip(tw.names, gs)]
            return (None, None, None, None, None, None, gx, *gal, *gpar)

    _PaddedNetFn = PaddedNetFn
    return _PaddedNetFn


_PaddedNetFn = None


def run_padded(net, x, alphas=None, loss_target=None, smooth=1e-6):
    """forward (probabilities, or (Dice loss, probabilities) with loss_target) of a net with odd channel counts through its padded twin"""
    import torch
    tw = net.__dict__.get("_n3d_twin")
    if tw is None or next(net.parameters()).device != next(tw.twin.parameters()).device:
        tw = net.__dict__["_n3d_twin"] = net._n3d_make_twin()      # (kept out of the module tree: the state dict stays the reference's)
    rp = dict(net.named_parameters())
    reals = [rp[n] for n in tw.names]
    als = tuple(alphas) if alphas is not None else ()
    # inference (no_grad / nothing requires a gradient): the twin runs under no_grad too -- no autograd graph, no saved activations
    need_grad = torch.is_grad_enabled() and (x.requires_grad or any(r.requires_grad for r in reals) or any(a.requires_grad for a in als))
    res = _padded_net_fn().apply(net, tw, need_grad, loss_target, smooth, len(als), x, *als, *reals)
    return res[0] if loss_target is None else res


def build_stems_and_head(net, in_channels, init_n_kernels, out_channels, n_nodes, head_in, head_dropout):
    """registers stem0 / stem1 / last_conv on `net` under the reference's attribute names (state-dict parity)"""
    wide = n_nodes * init_n_kernels
    net.stem0 = ConvOps(in_channels, wide, kernel_size=1, ops_order="weight_norm")
    net.stem1 = ConvOps(in_channels, wide, kernel_size=3, stride=2, ops_order="weight_norm")
    return nn.Sequential(ConvOps(head_in, out_channels, kernel_size=1, dropout_rate=head_dropout, ops_order="weight"), nn.Sigmoid())


def set_storage(net, storage):
    """"fp32" (the reference's arithmetic, default) or "bf16": bf16 STORAGE of the activations and activation gradients of
    the stems and of the cells with node width <= fused.BF16_MAX_NODE_WIDTH (the HBM-bound levels; BASELINE configs[4]: 4x128^3
    patches), fp32 arithmetic everywhere, fp32 weights / statistics / deep levels.  Searched nets only (the supernet's N-term
    kernels are fp32)."""
    if storage not in ("fp32", "bf16"):
        raise ValueError("storage must be 'fp32' or 'bf16'")
    net._n3d_storage = storage
    net._net_plan = None


def body(net, x, alphas=None, planar=False):
    """Everything ahead of the head.  alphas: None (searched net) or (alpha1_down, alpha1_up, alpha2_down, alpha2_up), already
    softmaxed.  Stems and cells run as one autograd node (fused.NetFn) unless fused.WHOLE_NET is off.
    planar: the caller hands the result to the fused head and nothing else -- the last cell may then keep its node outputs as
    dense tensors (a 6-D node-planar result, fused.PLANAR_LAST) instead of one concatenation buffer."""
    if not fused.WHOLE_NET:
        if alphas is None:
            plain = lambda cell, skip, cur: cell(skip, cur)
            return route(net, x, plain, plain, head=False)
        a1d, a1u, a2d, a2u = alphas
        return route(net, x, lambda cell, skip, cur: cell(skip, cur, a1d, a2d), lambda cell, skip, cur: cell(skip, cur, a1u, a2u), head=False)
    plan = getattr(net, "_net_plan", None)
    if not fused.current(plan):      # first call, or an op's norm / dropout / conv was re-assigned since
        plan = net._net_plan = fused.net_plan(net, supernet=alphas is not None)
    prev, fused.PLANAR_OUT = fused.PLANAR_OUT, bool(planar and fused.PLANAR_LAST)
    try:
        return fused.NetFn.apply(plan, x, *(alphas if alphas is not None else (None,) * 4), *plan.params)
    finally:
        fused.PLANAR_OUT = prev


def run(net, x, alphas=None):
    """Forward of either net: probabilities (B, n_out, D, H, W).  The head (Dropout3d -> 1x1x1 conv -> sigmoid) is one launch."""
    return _head.run(net.last_conv, body(net, x, alphas, planar=_head_takes_planar(net)))


def run_loss(net, x, t, alphas=None, smooth=1e-6):
    """(Dice loss, probabilities): forward with the loss of loss.py:12-14 formed inside the head's passes (the trainers' path)"""
    import torch
    return _head.run_loss(net.last_conv, body(net, x, alphas, planar=_head_takes_planar(net) and t.dtype == torch.float32), t, smooth)


def _head_takes_planar(net):
    """is the head the fused kernel pair (which reads a node-planar input), not the op-by-op fallback?"""
    import torch
    ok = getattr(net, "_n3d_head_planar", None)
    if ok is None:
        ci = net.last_conv[0].conv.weight.shape[1] if hasattr(net.last_conv[0], "conv") else 0
        ok = net._n3d_head_planar = bool(ci) and _head.fusable(net.last_conv, torch.empty((1, ci, 1, 1, 1), device="meta"))
    return ok


def route(net, x, call_down, call_up, head=True):
    """stems -> down cells (every output is kept as a skip) -> up cells (each takes the latest remaining skip) [-> head]"""
    older, newer = net.stem0(x), net.stem1(x)
    kept = [older, newer]
    for cell in net.down_cells:
        older, newer = newer, call_down(cell, older, newer)
        kept.append(newer)
    kept.pop()
    for cell in net.up_cells:
        newer = call_up(cell, kept.pop(), newer)
    return net.last_conv(newer) if head else newer
