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
    widths = sorted({wide} | {spec[2] for spec in specs})
    if any(c % 4 for c in widths):
        # the kernels move channels four at a time (16-byte NDHWC vectors); only a conv INPUT may have another count
        raise NotImplementedError("nas_3d_unet_amd: every feature-map channel count must be a multiple of 4; init_n_kernels=%d, n_nodes=%d, "
                                  "depth=%d give stems of %d and cell nodes of %s channels (config.yml's 4 / 3 / 4: 12 and 4..64)"
                                  % (init_n_kernels, n_nodes, depth, wide, sorted({spec[2] for spec in specs})))
    return specs, prev1


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
