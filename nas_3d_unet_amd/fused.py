"""Cell-level autograd Functions: one autograd node per MixedOp / Cell / SearchedCell.

The reference builds a cell out of Python-level tensor algebra -- ``w * op(x)``, ``sum(...)``,
``torch.cat(dim=1)`` (cell.py:29-32,76-82; searched.py:45-51) -- which costs one extra full pass
(and one launch) per term.  Here the whole cell is one launch program:
  * every primitive's normalise / ReLU / alpha-weight epilogue ACCUMULATES straight into its node's
    channel slice of the cell output buffer (zero-copy concat, fused weighted sum);
  * backward walks the DAG in reverse, data-gradient kernels accumulate into the producers'
    gradient slices, and dalpha[e][k] = <d node, op_k(x)> falls out of the epilogue's reduction pass.
Numerics: a searched cell accumulates in the reference's left-to-right order; a supernet node in the fixed order of
_node_units (terms grouped so that launches can be shared), i.e. the same sum with a different fp32 rounding order.
"""
from __future__ import annotations


import torch

from . import kernels as K
from . import programs as P
from ._lib import ACCUMULATE


PAIR_SUPERNET_TERMS = True  # pair the GroupNorm-type terms of a supernet node (False: one epilogue launch per primitive)
GROUP_SUPERNET_TERMS = True  # ... and take them up to eight at a time where a node has three or more (P.group_forward)
REUSE_GRAD_OUTPUT = False  # see _run_backward; switched on by the trainers for the duration of their backward pass


def _grouping(c_node):
    """are the N-term launches of a supernet node in use?  (both switches on, channel count the group kernels take)"""
    return PAIR_SUPERNET_TERMS and GROUP_SUPERNET_TERMS and K.group_shape_ok(c_node)


def _single_segment(op):
    if op._segments is None:
        op._segments = op._build_segments()
    if len(op._segments) != 1:
        raise P.N3DError("fused cells need single-segment primitives, got ops_order=%s" % "_".join(op.ops_list))
    return op._segments[0]


def _slice_view(buf_view, k, cn):
    """View of channels [k*cn, (k+1)*cn) of a (B, n*cn, D, H, W) NDHWC buffer."""
    t = buf_view.t[:, k * cn:(k + 1) * cn]
    return K.View(t, buf_view.ld)


def _copy_into(src, dst):
    """dst = src through the identity epilogue kernel (both pitched views)."""
    K.affine_act(src, None, None, None, dst, 0)


class _Plan:
    """Static description of a cell: edges = (node, input index, [(segment, alpha_col)], alpha matrix id, row)."""

    def __init__(self, pre0, pre1, n_nodes, c_node, edges, params, pairs=False):
        self.pre0, self.pre1, self.n_nodes, self.c_node, self.edges, self.params = pre0, pre1, n_nodes, c_node, edges, params
        self.index = {id(p): i for i, p in enumerate(params)}
        self.pairs = pairs  # searched cell: node k = exactly two single-primitive edges (2k, 2k+1)
        self.version = P.PLAN_VERSION[0]   # of the ops' launch programs this plan was built from
        self.dt = torch.float32  # storage type of the cell's activations (bf16 configuration: set by _NetPlan)


def searched_plan(cell):
    edges = []
    for node in range(cell.n_nodes):
        for e in (2 * node, 2 * node + 1):
            edges.append((node, cell.genolist[e][1], [(_single_segment(cell._ops[e]), 0)], 0, e))
    return _Plan(_single_segment(cell.preprocess0), _single_segment(cell.preprocess1), cell.n_nodes, cell.c_node, edges,
                 list(cell.parameters()), pairs=True)


def supernet_plan(cell):
    edges = []
    e = 0
    for node in range(cell.n_nodes):
        for i in range(node + 2):
            mixed = cell._ops[e]
            segs = [(_single_segment(op), k) for k, op in enumerate(mixed._ops)]
            edges.append((node, i, segs, 1 if mixed.stride == 1 else 2, e))
            e += 1
    return _Plan(_single_segment(cell.preprocess0), _single_segment(cell.preprocess1), cell.n_nodes, cell.c_node, edges,
                 list(cell.parameters()))


def _flat_terms(plan):
    """(node, input index, segment, alpha column, alpha matrix id, alpha row) of every primitive, in forward order"""
    flat = []
    for node, idx, segs, amat, row in plan.edges:
        for seg, col in segs:
            flat.append((node, idx, seg, col, amat, row))
    return flat


def _node_units(plan):
    """Launch schedule of a supernet cell: per node a list of units, each one term index or a pair of term indices into
    _flat_terms(plan).  GroupNorm-type terms share their epilogue launches two at a time (P.pair_forward / pair_backward).
    The plain-conv terms are paired with each other, partners taken from DIFFERENT edges where the node has more than one:
    then the two convs also share one launch forward (n3d_conv_fwd2) and, having distinct input-gradient targets, one
    launch backward (n3d_conv_bwd_both2).  The remaining GroupNorm-type terms (depthwise-separable, stride-2 SE convs)
    pair in order; identity, pooling and stride-1 SE terms stay single."""
    if getattr(plan, "_units", None) is not None:
        return plan._units
    flat = _flat_terms(plan)
    units = [_units_of(plan, flat, [fi for fi, t in enumerate(flat) if t[0] == node]) for node in range(plan.n_nodes)]
    plan._units = units
    return units


def _node_units_split(plan, side_inputs=(0, 1)):
    """the units of each node for the two-stream BACKWARD (SIDE_BWD): (units of the main stream's terms, units of the side stream's).
    The side stream takes the terms whose input is one of the preprocess outputs in `side_inputs` (both, or only the second one
    when it has other work); the main stream keeps the terms that read a NODE -- their input gradients are what the next node's
    backward waits for -- and the rest.  The two sets accumulate their input gradients into disjoint tensors (a preprocess
    gradient belongs to ONE stream), so they can run on two streams without ordering between them."""
    cache = plan.__dict__.setdefault("_units_split", {})
    key = tuple(side_inputs)
    if key in cache:
        return cache[key]
    flat = _flat_terms(plan)
    out = []
    for node in range(plan.n_nodes):
        mine = [fi for fi, t in enumerate(flat) if t[0] == node]
        out.append((_units_of(plan, flat, [fi for fi in mine if flat[fi][1] not in key]), _units_of(plan, flat, [fi for fi in mine if flat[fi][1] in key])))
    cache[key] = out
    return out


def _units_of(plan, flat, mine):
    """launch units of the terms `mine` (indices into flat) of one node: see _node_units"""
    if True:
        pairable = [fi for fi in mine if PAIR_SUPERNET_TERMS and P.gn_pairable(flat[fi][2])]
        dense = [fi for fi in pairable if isinstance(flat[fi][2].weight, P.DenseConvW)]
        # position of the term among its edge's dense terms first, edge second: neighbours come from different edges
        rank, seen = {}, {}
        for fi in dense:
            e = flat[fi][5]
            rank[fi] = seen.get(e, 0)
            seen[e] = rank[fi] + 1
        dense.sort(key=lambda fi: (rank[fi], fi))
        rest = [fi for fi in pairable if fi not in rank]
        rest.sort(key=lambda fi: (not isinstance(flat[fi][2].weight, P.DepthSepW), fi))   # depthwise-separable terms side by side
        u = []
        if _grouping(plan.c_node) and len(pairable) >= 3 and len({flat[fi][2].norm.eps for fi in pairable}) == 1:
            # N-term groups: all coefficients in one launch and one pass over the node buffer (backward: three launches per group)
            both = dense + rest
            for i in range(0, len(both), K.MAX_GROUP_TERMS):
                u.append(tuple(both[i:i + K.MAX_GROUP_TERMS]))
        else:
            for group in (dense, rest):
                for i in range(0, len(group) - 1, 2):
                    u.append((group[i], group[i + 1]))
                if len(group) % 2:
                    u.append((group[-1],))
        u.extend((fi,) for fi in mine if fi not in pairable)
        return u


def _node_fwd_units(plan):
    """Forward launch schedule of a supernet cell.  With N-term groups the forward pass of a node takes ALL its primitives up
    to eight at a time (P.group_forward: the pass over the node buffer does not care what produced a term's scale / shift),
    in the order of _node_units followed by the primitives that stay single in backward."""
    if getattr(plan, "_fwd_units", None) is not None:
        return plan._fwd_units
    units = _node_units(plan)
    if _grouping(plan.c_node):
        out = []
        for u in units:
            every = [fi for unit in u for fi in unit]
            out.append([tuple(every[i:i + K.MAX_GROUP_TERMS]) for i in range(0, len(every), K.MAX_GROUP_TERMS)])
        units = out
    plan._fwd_units = units
    return units


def _run_forward(plan, x0, x1, alpha1, alpha2, pre0_early=None, planar=False):
    """Returns (cell output tensor, saved state).  pre0_early = (View, saved state, flag id): the cell's first preprocess op has
    already been launched on the side stream (NetFn.forward, SIDE_FWD); the cell joins it instead of running it.
    planar (searched cells): the node outputs stay dense tensors (kernels.Planar, a 6-D output) instead of channel slices of one
    concatenation buffer -- for the net's LAST cell, whose only reader is the fused head."""
    with K.stats_cache(), K.storage(plan.dt, getattr(plan, "mm_bf16", False)):
        return _run_forward_impl(plan, x0, x1, alpha1, alpha2, pre0_early, planar)


# The last cell's output feeds the head only (nas.py:77-78, searched.py:110-111).  As channel slices of one (B, 3 c) buffer every
# node epilogue writes 16 bytes on a 48-byte pitch and every epilogue backward reads its gradient the same way: 0.28-0.47 of the HBM
# roofline at 4x128^3 (round-3 review).  With PLANAR_LAST the callers that own both sides (unet.body -> head.run / run_loss, the
# trainers' pipeline) set PLANAR_OUT around NetFn.forward and the last cell keeps its nodes dense; the head gathers three pointers.
PLANAR_LAST = True     # (tests switch it off to compare with the concatenated layout)
PLANAR_OUT = False
# Round 5: the same for every cell whose concatenation is read by ONE 1x1x1 preprocess conv of the next cell and nothing else (searched
# nets: the up cells -- an up cell's output is only the next up cell's x1) on >= 32768 voxels: the conv takes the node-planar tensor as it
# is (include/n3d.h, "node-planar tensors": pitch < channels), its data gradient writes the per-node gradients, its weight gradient gathers
# the nodes.  Here that is up-cell 3 (3 x 8 channels at 32^3 / 64^3 for 64^3 / 128^3 patches).
PLANAR_INNER = True    # (tests switch it off to compare with the concatenated layout)


# Side-stream forward of a supernet cell (train.SideSchedule sets SIDE_FWD; round 3).  A node sums the MixedOps of ALL earlier states
# (cell.py:76-81), but only the edge from the node computed LAST is on the dependent chain: the weight ops of every other edge read
# tensors that have been complete for a while.  With SIDE_FWD set, those run on the side stream -- the edges from the two preprocess
# outputs into nodes 1.. and one of node 0's two edges right behind the preprocess ops, an edge from node k into a later node as
# soon as node k is written -- and the main stream keeps one edge per node plus the epilogues, which wait (device flag) for the
# side stream's part of their node.  SIDE_FWD: .fork() main stores a flag, returns its id; .side(wait_id) context manager: launches
# go to the side stream behind a wait on that flag; .side_signal() side stores a flag, returns its id; .join(id) main waits for it.
SIDE_FWD = None
NODE_APPLY = True     # ... and one apply launch for the level's SE gates and identity primitives (tests compare with the per-term launches)
NODE_PHASES = True    # one reduction + one coefficient launch per node level of the supernet backward (tests compare with the per-term launches)
SIDE_BWD = None      # the same object while the backward pass of a supernet may use the side stream (see _run_backward_impl)


def _node_buffer(plan, shape, device, planar):
    """(output object, node Views): one (B, n c) concatenation buffer with the nodes as channel slices, or -- planar -- the nodes as
    dense tensors of their own (kernels.Planar).  planar: True (the head reads it) or the output channel count of the ONE 1x1x1
    preprocess conv that reads it (the node-planar form is taken where the streaming 1x1x1 kernels take it: channel counts, voxels)"""
    cn, nn = plan.c_node, plan.n_nodes
    if planar is not True and planar:
        co = int(planar)
        vox = shape[2] * shape[3] * shape[4]
        planar = (vox >= 32768 and cn % 4 == 0 and cn <= 12 and nn * cn <= 24 and co % 4 == 0 and co <= 12
                  and (nn * cn // 4) * (co // 4) <= 6 and (co // 4) * (cn // 4) <= 6)
    if planar:
        out = K.empty_planar(nn, shape[0], cn, shape[2], shape[3], shape[4], device)
        return out, out.nodes
    out = K.as_view(K.empty_ndhwc(shape[0], nn * cn, shape[2], shape[3], shape[4], device))
    return out, [_slice_view(out, k, cn) for k in range(nn)]


def _run_forward_side(plan, x0, x1, alpha1, alpha2, sf, planar=False):
    st = P.Saved()
    x0v, x1v = K.as_view(x0, "x0"), K.as_view(x1, "x1")
    shp = plan.pre1.weight.out_shape(x1v)
    p0 = K.as_view(K.empty_ndhwc(*shp, x1v.t.device))
    p1 = K.as_view(K.empty_ndhwc(*shp, x1v.t.device))
    if tuple(plan.pre0.weight.out_shape(x0v)) == tuple(shp) and plan.pre0.dropout is None and plan.pre1.dropout is None:
        st.s_pre0, st.s_pre1 = P.pair_forward(plan.pre0, x0v, plan.pre1, x1v, p0, p1)
    else:
        p0, st.s_pre0 = P.seg_forward(plan.pre0, x0v)
        p1, st.s_pre1 = P.seg_forward(plan.pre1, x1v)
    xs = [p0, p1]
    cn, nn = plan.c_node, plan.n_nodes
    flat = _flat_terms(plan)
    units = _node_fwd_units(plan)
    # the node buffer (its shape is that of any term's output)
    seg0, in0 = flat[0][2], xs[flat[0][1]]
    oshp = seg0.weight.out_shape(in0)
    out, nodes = _node_buffer(plan, oshp, in0.t.device, planar)
    xs.extend(nodes)

    def args_of(fi):
        _, idx, seg, col, amat, row = flat[fi]
        arow = (alpha1 if amat == 1 else alpha2)[row] if amat else None
        return (seg, xs[idx], arow, col)

    def on_side(node, idx):
        """does the weight phase of a term of `node` reading state `idx` run on the side stream?"""
        if node == 0:
            return False            # node 0: both edges on the main stream (one of them on the side stream measured 0.1 ms slower per search step)
        return idx <= node          # node n's newest input is state n + 1

    order = {node: [fi for unit in units[node] for fi in unit] for node in range(nn)}     # the epilogue order of the node's terms
    res = {}
    side_done = {}
    # side stream, behind the preprocess ops: every term that reads p0 / p1 and is not main's
    f0 = sf.fork()
    with sf.side(f0):
        for node in range(nn):
            mine = [fi for fi in order[node] if flat[fi][1] <= 1 and on_side(node, flat[fi][1])]
            if mine:
                for fi, r in zip(mine, P.group_weight_phase([args_of(fi) for fi in mine])):
                    res[fi] = r
                side_done.setdefault(node, []).append(sf.side_signal())
    st.saved = [None] * len(flat)
    for node in range(nn):
        mine = [fi for fi in order[node] if not on_side(node, flat[fi][1])]
        for fi, r in zip(mine, P.group_weight_phase([args_of(fi) for fi in mine])):
            res[fi] = r
        for tok in side_done.get(node, []):
            sf.join(tok)
        started = False
        for unit in units[node]:
            for fi, sv in zip(unit, P.group_epilogue_phase([args_of(fi) for fi in unit], [res[fi] for fi in unit], nodes[node], started)):
                st.saved[fi] = sv
            started = True
        # node `node` is written: the side stream may take the edges from it into the nodes after the next one
        later = [(n2, fi) for n2 in range(node + 2, nn) for fi in order[n2] if flat[fi][1] == node + 2]
        if later:
            f = sf.fork()
            with sf.side(f):
                for n2 in sorted({n2 for n2, _ in later}):
                    mine = [fi for m, fi in later if m == n2]
                    for fi, r in zip(mine, P.group_weight_phase([args_of(fi) for fi in mine])):
                        res[fi] = r
                    side_done.setdefault(n2, []).append(sf.side_signal())
    st.xs, st.out = xs, out
    return out.t, st


SIDE_PAIRS = True   # searched cells with C <= 8: one conv of a node early on the side stream
SIDE_STEM1_BWD = True   # stem1's backward (parameter gradients only) beside stem0's
SIDE_PRE0_BWD = True     # searched cells: the backward of EVERY cell's first preprocess op on the side stream
SIDE_PAIRS_BOTH = True   # ... and of nodes whose two convs both read node outputs, one conv beside the other (1.756 -> 1.748 ms)
SIDE_PAIRS_BWD = True   # ... and the data gradients into one preprocess gradient (needs a third stream)


def _early_pair_ops(plan):
    """{node: edge index} -- for every node of a searched cell at most ONE op that reads a preprocess output (state 0 / 1) and is a plain
    conv: the op the side stream runs ahead of the node chain.  Of two such ops the transposed / strided one goes (the cheaper launch)."""
    got = getattr(plan, "_early_pairs", None)
    if got is not None:
        return got
    got = {}
    for node in range(plan.n_nodes):
        cands = []
        for e in (2 * node, 2 * node + 1):
            _, idx, segs, _, _ = plan.edges[e]
            seg = segs[0][0]
            if idx <= 1 and isinstance(seg.weight, P.DenseConvW) and P.gn_pairable(seg):
                cands.append((0 if (seg.weight.transposed or seg.weight.stride > 1) else 1, e))
        # the other op of the node must be a pair-epilogue op too (pair_epilogue_phase)
        other_ok = all(P.gn_pairable(plan.edges[e][2][0][0]) for e in (2 * node, 2 * node + 1))
        if cands and other_ok:
            got[node] = sorted(cands)[0][1]
    plan._early_pairs = got
    return got


def _run_forward_impl(plan, x0, x1, alpha1, alpha2, pre0_early=None, planar=False):
    if SIDE_FWD is not None and not plan.pairs and _grouping(plan.c_node):
        return _run_forward_side(plan, x0, x1, alpha1, alpha2, SIDE_FWD, planar)
    st = P.Saved()
    x0v, x1v = K.as_view(x0, "x0"), K.as_act(x1, "x1")     # (x1 may be the previous up cell's node-planar output)
    # the two preprocess ops (cell.py:47-50) are independent and of one output shape: paired epilogue launch
    shp = plan.pre1.weight.out_shape(x1v)
    if pre0_early is not None:
        # the skip-side preprocess op ran on the side stream as soon as its input existed: only the other one is on the chain
        p0, st.s_pre0, tok = pre0_early
        p1, st.s_pre1 = P.seg_forward(plan.pre1, x1v)
        SIDE_FWD.join(tok)
        st.pre0_side = True
    elif tuple(plan.pre0.weight.out_shape(x0v)) == tuple(shp) and plan.pre0.dropout is None and plan.pre1.dropout is None:
        p0 = K.as_view(K.empty_ndhwc(*shp, x1v.t.device))
        p1 = K.as_view(K.empty_ndhwc(*shp, x1v.t.device))
        st.s_pre0, st.s_pre1 = P.pair_forward(plan.pre0, x0v, plan.pre1, x1v, p0, p1)
    else:
        p0, st.s_pre0 = P.seg_forward(plan.pre0, x0v)
        p1, st.s_pre1 = P.seg_forward(plan.pre1, x1v)
    xs = [p0, p1]
    cn, nn = plan.c_node, plan.n_nodes
    out = None
    started = [False] * nn
    st.saved = []
    if plan.pairs:
        # searched cell: each node is one pair (both weight ops, then ONE epilogue launch writing the node slice)
        early = _early_pair_ops(plan) if (SIDE_FWD is not None and SIDE_PAIRS and cn <= 8) else {}
        side_res = {}
        if early:
            # C <= 8: the two convs of a node are two launches (no common MFMA problem).  The ones that read a preprocess output do
            # not depend on the node chain: the side stream takes them all now, in node order, a flag behind each
            sf = SIDE_FWD
            f0 = sf.fork()
            with sf.side(f0):
                for node in sorted(early):
                    e = early[node]
                    _, idx, segs, _, _ = plan.edges[e]
                    side_res[node] = (e, P.pair_weight_phase(segs[0][0], xs[idx]), sf.side_signal())
        for node in range(nn):
            (_, i0, segs0, _, _), (_, i1, segs1, _, _) = plan.edges[2 * node], plan.edges[2 * node + 1]
            seg0, seg1 = segs0[0][0], segs1[0][0]
            if out is None:
                out, nodes = _node_buffer(plan, seg0.weight.out_shape(xs[i0]), xs[i0].t.device, planar)
                xs.extend(nodes)
            if node in side_res:
                e, res_side, tok = side_res[node]
                if e == 2 * node:
                    res0, res1 = res_side, P.pair_weight_phase(seg1, xs[i1])
                else:
                    res0, res1 = P.pair_weight_phase(seg0, xs[i0]), res_side
                SIDE_FWD.join(tok)
                s0, s1 = P.pair_epilogue_phase(seg0, res0, seg1, res1, nodes[node])
            elif (SIDE_PAIRS_BOTH and SIDE_FWD is not None and cn <= 8 and isinstance(seg0.weight, P.DenseConvW)
                  and isinstance(seg1.weight, P.DenseConvW) and P.gn_pairable(seg0) and P.gn_pairable(seg1)):
                # both convs read node outputs: still two launches (no common MFMA problem) -- the cheaper one beside the other
                sf = SIDE_FWD
                cheap1 = seg1.weight.transposed or seg1.weight.stride > 1
                sa, ia, sb_, ib = (seg1, i1, seg0, i0) if cheap1 else (seg0, i0, seg1, i1)
                with sf.side(sf.fork()):
                    res_a = P.pair_weight_phase(sa, xs[ia])
                    tok = sf.side_signal()
                res_b = P.pair_weight_phase(sb_, xs[ib])
                sf.join(tok)
                res0, res1 = (res_b, res_a) if cheap1 else (res_a, res_b)
                s0, s1 = P.pair_epilogue_phase(seg0, res0, seg1, res1, nodes[node])
            else:
                s0, s1 = P.pair_forward(seg0, xs[i0], seg1, xs[i1], nodes[node])
            st.saved.extend([s0, s1])
        st.xs, st.out = xs, out
        return out.t, st
    # supernet cell (cell.py:76-81): node = sum over its edges of sum_k alpha[e][k] * op_k(x_e), accumulated unit by unit
    # in the order of _node_units (a fixed order, but not the reference's left-to-right one: fp32 rounding differs)
    flat = _flat_terms(plan)
    st.saved = [None] * len(flat)
    grouped = _grouping(plan.c_node)
    for node, units in enumerate(_node_fwd_units(plan)):
        for unit in units:
            args = []
            for fi in unit:
                _, idx, seg, col, amat, row = flat[fi]
                xin = xs[idx]
                if out is None:
                    out, nodes = _node_buffer(plan, seg.weight.out_shape(xin), xin.t.device, planar)
                    xs.extend(nodes)
                arow = (alpha1 if amat == 1 else alpha2)[row] if amat else None
                args.append((seg, xin, arow, col))
            if grouped and len(unit) >= 2:
                for fi, sv in zip(unit, P.group_forward(args, nodes[node], started[node])):
                    st.saved[fi] = sv
            elif len(unit) == 2:
                (sa_, xa, aa, ca), (sb_, xb, ab, cb) = args
                st.saved[unit[0]], st.saved[unit[1]] = P.pair_forward(sa_, xa, sb_, xb, nodes[node], None, started[node], (aa, ca), (ab, cb))
            else:
                (sa_, xa, aa, ca), = args
                _, st.saved[unit[0]] = P.seg_forward(sa_, xa, None, nodes[node], started[node], aa, ca)
            started[node] = True
    st.xs, st.out = xs, out
    return out.t, st


def _run_backward(plan, st, dout, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets=None, own_dout=False, late_joins=()):
    """dx_targets = ((View | None, accumulate), (View | None, accumulate)): where the gradients of the two cell inputs go
    (NetFn passes the producers' gradient buffers); own_dout: `dout` is a private buffer the cell may accumulate into;
    late_joins: flags of side-stream work that is still writing one of the two target buffers -- joined in front of the preprocess
    backward, the first thing of this cell that touches them."""
    with K.storage(plan.dt, getattr(plan, "mm_bf16", False)):
        if SIDE_BWD is not None and not plan.pairs:
            with SIDE_BWD.arena_mode():      # (this runs on autograd's thread: the mode of the forward pass is not active here)
                return _run_backward_impl(plan, st, dout, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, own_dout, late_joins)
        return _run_backward_impl(plan, st, dout, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, own_dout, late_joins)


def _run_backward_impl(plan, st, dout, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, own_dout, late_joins=()):
    cn, nn = plan.c_node, plan.n_nodes
    out = st.out
    dev = out.t.device
    if isinstance(out, K.Planar):
        # node-planar output (the net's last cell): the gradient arrives as dense per-node tensors too
        dv = K.as_planar(dout, "grad_output")
        if not ((REUSE_GRAD_OUTPUT or own_dout) and dv.t is dout):
            dcat = K.empty_planar(out.nn, out.B, out.cn, out.D, out.H, out.W, dev, out.t.dtype)
            dcat.t.copy_(dv.t)
            dv = dcat
        return _run_backward_nodes(plan, st, dv.nodes, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, late_joins)
    dv = K.as_view(dout, "grad_output")
    # node slices of the incoming gradient receive further contributions.  A trainer that owns the whole backward
    # (train.Trainer / SearchTrainer set REUSE_GRAD_OUTPUT) lets the cell accumulate straight into autograd's buffer: it is
    # either this cell's consumer's freshly returned dx or autograd's own accumulation buffer, and nobody reads it again.
    # Stand-alone use keeps the private copy (autograd forbids modifying grad_output in general).
    if (REUSE_GRAD_OUTPUT or own_dout) and dv.t is dout and dv.ld == nn * cn:
        dcat = dv
    else:
        dcat = K.like(out)
        _copy_into(dv, dcat)
    dnodes = [_slice_view(dcat, k, cn) for k in range(nn)]
    return _run_backward_nodes(plan, st, dnodes, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, late_joins)


def _run_backward_nodes(plan, st, dnodes, alpha1, alpha2, need_x0, need_x1, want_dalpha, dx_targets, late_joins=()):
    """the backward walk of a cell given the gradient Views of its nodes (channel slices of one buffer, or dense tensors)"""
    cn, nn = plan.c_node, plan.n_nodes
    dev = st.out.t.device
    p0, p1 = st.xs[0], st.xs[1]
    dpre = [K.like(p0), K.like(p1)]
    pre_started = [False, False]
    grads = [None] * len(plan.params)
    da1 = torch.zeros_like(alpha1) if (want_dalpha and alpha1 is not None) else None
    da2 = torch.zeros_like(alpha2) if (want_dalpha and alpha2 is not None) else None

    def put(seg, glist):
        for p, g in zip(seg.params(), glist):
            if g is not None and getattr(p, "_n3d_grad", None) is None:
                grads[plan.index[id(p)]] = g

    def tgt(idx):
        if idx >= 2:
            return dnodes[idx - 2], True
        t, acc = dpre[idx], pre_started[idx]
        pre_started[idx] = True
        return t, acc

    if plan.pairs:
        # searched cell: nodes in reverse; each node's two epilogue backwards share their launches (P.pair_backward)
        # C <= 8 (two data-gradient launches per node) with a weight-gradient stream of its own: every data gradient into ONE of
        # the two preprocess gradients -- the one with more writers -- runs on the side stream, behind a flag per node; the chain
        # keeps the gradients into the nodes (and into the other preprocess output) and joins once, in front of the preprocess
        # backward.  All writers of a buffer stay on one stream, in program order: same accumulation order, same bits.
        sb = SIDE_BWD
        side_idx, side_tok = None, None
        if sb is not None and SIDE_PAIRS_BWD and cn <= 8 and getattr(sb, "split", False):
            writers = [sum(1 for e in plan.edges if e[1] == k) for k in (0, 1)]
            side_idx = 0 if writers[0] > writers[1] else 1
        for node in reversed(range(nn)):
            (_, i0, segs0, _, _), (_, i1, segs1, _, _) = plan.edges[2 * node], plan.edges[2 * node + 1]
            seg0, seg1 = segs0[0][0], segs1[0][0]
            s0, s1 = st.saved[2 * node], st.saved[2 * node + 1]
            t1, a1 = tgt(i1)   # second edge first: same accumulation order as the unpaired reverse walk
            t0, a0 = tgt(i0)
            items = None
            if side_idx is not None and side_idx in (i0, i1):
                items = P.pair_backward_epilogue(seg0, s0, seg1, s1, dnodes[node], (True, t0, a0), (True, t1, a1))
            if items is None:
                (_, g0), (_, g1) = P.pair_backward(seg0, s0, seg1, s1, dnodes[node], (True, t0, a0), (True, t1, a1))
            else:
                it0, it1 = items
                res = {}
                on_side = [(1, it1)] if i1 == side_idx else []
                if i0 == side_idx:
                    on_side.append((0, it0))
                on_main = [(k, it) for k, it, idx in ((1, it1, i1), (0, it0, i0)) if idx != side_idx]
                ready = sb.fork()
                with sb.side(ready):
                    for (k, _), r in zip(on_side, P._weight_backward(tuple(it for _, it in on_side))):
                        res[k] = r
                    side_tok = sb.side_signal()
                if on_main:
                    for (k, _), r in zip(on_main, P._weight_backward(tuple(it for _, it in on_main))):
                        res[k] = r
                g0, g1 = res[0][1], res[1][1]
            put(seg1, g1)
            put(seg0, g0)
            if NODE_DONE_HOOK is not None:
                NODE_DONE_HOOK()
        if side_tok is not None:
            late_joins = [side_tok] + list(late_joins)     # joined with the late writers below: two flags per wait launch
        flat = []
    else:
        flat = None
    # supernet cell: reverse order over primitives (saved states are in forward order); the GroupNorm-type terms of a
    # node share their epilogue-backward launches two at a time (they all consume the same node gradient)
    if flat is None:
        flat = _flat_terms(plan)

        def alpha_of(amat, row):
            if not amat:
                return None, None
            d = da1 if amat == 1 else da2
            return (alpha1 if amat == 1 else alpha2)[row], (d[row] if d is not None else None)

        all_units = _node_units(plan)
        batch_reduce = _grouping(cn)

        def run_node(node, node_units):
            """backward of the terms in `node_units` (units of ONE node, all of them or the subset one stream handles)"""
            # the primitives that stay single (identity, SE gates, pooling) all start with a reduction pass over the same node
            # gradient: those passes run up to eight per launch
            pre, pre_se, pre_da, prep, done = {}, {}, set(), {}, {}
            if batch_reduce:
                want = []
                for unit in node_units:
                    if len(unit) == 1:
                        fi, = unit
                        _, _, seg, _, amat, row = flat[fi]
                        s = st.saved[fi]
                        if P.needs_reduce(seg, s, alpha_of(amat, row)[1]):
                            want.append((fi, (s.raw, s.a if s.kind != "plain" else None, s.b if s.kind == "gn" else None,
                                              seg.relu_out and s.kind != "se")))
                # NODE_PHASES (round 4): with N-term GroupNorm groups among the units, the node level's reductions are ONE launch (the
                # groups' terms and the single primitives', up to 16) and its coefficient computations another (GroupNorm terms and SE
                # gates behind a per-workgroup switch) -- P.group_backward then only runs the apply pass and the weight ops
                group_units = [u for u in node_units if len(u) >= 3] if NODE_PHASES else []
                if group_units and NODE_APPLY:
                    # a leftover PAIR of GroupNorm terms (the ninth and tenth of a node) rides in the same phases instead of its own
                    # reduce2 + apply_gn2 launches
                    group_units += [u for u in node_units if len(u) == 2 and all(st.saved[fi].kind == "gn" and P.gn_pairable(flat[fi][2]) for fi in u)
                                    and st.saved[u[0]].G == st.saved[group_units[0][0]].G]
                if group_units:
                    for u in group_units:
                        tl = []
                        for fi in u:
                            _, _, seg, col, amat, row = flat[fi]
                            arow, dal = alpha_of(amat, row)
                            tl.append((seg, st.saved[fi], (arow, col, dal)))
                        prep[u] = P.group_prepare(tl, dnodes[node])
                    gate_specs = []
                    for k, (fi, _) in enumerate(want):
                        if st.saved[fi].kind == "se":
                            _, _, seg, col, amat, row = flat[fi]
                            arow, dal = alpha_of(amat, row)
                            s = st.saved[fi]
                            gate_specs.append((k, dict(wptr=P._wptr(arow, col), mean=s.mean, hidden=s.hidden, gate=s.a, fc=seg.se_gate.fc,
                                                       dalpha_ptr=(dal.data_ptr() + 4 * col) if dal is not None else None)))
                    ident_specs = []
                    if NODE_APPLY:
                        for k, (fi, _) in enumerate(want):
                            _, _, seg, col, amat, row = flat[fi]
                            s = st.saved[fi]
                            if s.kind == "gn" and isinstance(seg.weight, P.IdentityW) and s.G == prep[group_units[0]].G:
                                arow, dal = alpha_of(amat, row)
                                ident_specs.append((k, dict(gamma=seg.norm.weight, beta=seg.norm.bias, mr=s.mr, wptr=P._wptr(arow, col),
                                                            dalpha_ptr=(dal.data_ptr() + 4 * col) if dal is not None else None)))
                    rs, ses, ids = K.node_bwd_prologue(dnodes[node], [prep[u] for u in group_units], [c[1] for c in want], gate_specs, ident_specs)
                    for (fi, _), r in zip(want, rs):
                        pre[fi] = r
                    for (k, _), r in zip(gate_specs, ses):
                        pre_se[want[k][0]] = r
                    if NODE_APPLY:
                        # the apply passes of the SE gates and identity primitives: ONE launch, in the reverse walk's order per target
                        pre_id = {want[k][0]: r for (k, _), r in zip(ident_specs, ids) if r is not None}
                        items, outs = [], []
                        snap = list(pre_started)
                        for unit in reversed(node_units):
                            if len(unit) != 1 or (unit[0] not in pre_se and unit[0] not in pre_id):
                                continue
                            fi = unit[0]
                            _, idx, seg, col, amat, row = flat[fi]
                            s = st.saved[fi]
                            target, acc = tgt(idx)
                            if fi in pre_se:
                                dw1, db1, dw2, db2, A, Bc = pre_se[fi]
                                items.append((s.raw, None, None, False, A, Bc, None, target, acc))
                                outs.append((fi, seg, target, [dw1, db1, dw2, db2]))
                            else:
                                dgamma, dbeta, cA, cB, cC = pre_id[fi]
                                items.append((s.raw, s.a, s.b, seg.relu_out, cA, cB, cC, target, acc))
                                outs.append((fi, seg, target, [dgamma, dbeta]))
                        # (a target's terms must be consecutive for the launch: they are -- a target is one edge's input gradient and an
                        # edge's single primitives are neighbours in the walk; anything else falls back to the per-primitive path)
                        ok = len(items) >= 2 and len(items) <= K.MAX_REDUCE_TERMS
                        if ok:
                            seen, last = set(), None
                            per = {}
                            for it in items:
                                key = it[7].p.value
                                if key != last and key in seen:
                                    ok = False
                                seen.add(key); last = key
                                per[key] = per.get(key, 0) + 1
                            ok = ok and len(per) <= 8 and max(per.values()) <= 4
                        if ok:
                            K.node_bwd_apply_sum(dnodes[node], items)
                            for fi, seg, target, gl in outs:
                                done[fi] = (seg, gl)
                        else:
                            pre_started[:] = snap      # nothing was written: the per-primitive path claims the targets itself
                else:
                    for i in range(0, len(want), K.MAX_GROUP_TERMS):
                        chunk = want[i:i + K.MAX_GROUP_TERMS]
                        if len(chunk) >= 2:
                            for (fi, _), r in zip(chunk, K.affine_act_bwd_reduceN(dnodes[node], [c[1] for c in chunk])):
                                pre[fi] = r
                # ... the pooling primitives' dalpha = <d node, pooled> come out of one launch
                pools = [fi for fi, _ in want if fi in pre and isinstance(flat[fi][2].weight, P.PoolW) and not flat[fi][2].relu_out]
                for i in range(0, len(pools), K.MAX_GROUP_TERMS):
                    chunk = pools[i:i + K.MAX_GROUP_TERMS]
                    if len(chunk) >= 2:
                        tds = []
                        for fi in chunk:
                            _, _, _, col, amat, row = flat[fi]
                            dal = alpha_of(amat, row)[1]
                            tds.append((pre[fi][0], pre[fi][1], dal.data_ptr() + 4 * col))
                        raw = st.saved[chunk[0]].raw
                        K.plain_dalphaN(tds, raw.B, raw.C)
                        pre_da.update(chunk)
                # ... and the SE gates among them share one gate-backward launch
                gates = [fi for fi, _ in want if fi in pre and fi not in pre_se and st.saved[fi].kind == "se"]
                for i in range(0, len(gates), K.MAX_GROUP_TERMS):
                    chunk = gates[i:i + K.MAX_GROUP_TERMS]
                    if len(chunk) >= 2:
                        tds = []
                        for fi in chunk:
                            _, _, seg, col, amat, row = flat[fi]
                            arow, dal = alpha_of(amat, row)
                            s = st.saved[fi]
                            tds.append(dict(sums=pre[fi][0], rows=pre[fi][1], wptr=P._wptr(arow, col), mean=s.mean, hidden=s.hidden, gate=s.a,
                                            fc=seg.se_gate.fc, dalpha_ptr=(dal.data_ptr() + 4 * col) if dal is not None else None))
                        raw = st.saved[chunk[0]].raw
                        for fi, r in zip(chunk, K.se_gate_bwdN(tds, raw.N, raw.B, raw.C)):
                            pre_se[fi] = r
            rev = list(reversed(node_units))
            skip = set()
            for ui, unit in enumerate(rev):
                if ui in skip:
                    continue
                if len(unit) == 1 and unit[0] in done:      # its apply pass ran in the node level's merged launch (NODE_APPLY)
                    put(*done[unit[0]])
                    continue
                # the average and the max pooling of one edge (two single units next to each other): one pass over the input gradient
                if len(unit) == 1 and ui + 1 < len(rev) and len(rev[ui + 1]) == 1:
                    fb, fa = unit[0], rev[ui + 1][0]      # fb later in forward order
                    (_, ia, sega, cola, amata, rowa), (_, ib, segb, colb, amatb, rowb) = flat[fa], flat[fb]
                    if (isinstance(sega.weight, P.PoolW) and isinstance(segb.weight, P.PoolW) and ia == ib and sega.weight.is_max != segb.weight.is_max
                            and not (sega.relu_out or segb.relu_out)):
                        wps = {}
                        for fi, seg, col, amat, row in ((fa, sega, cola, amata, rowa), (fb, segb, colb, amatb, rowb)):
                            arow, dal = alpha_of(amat, row)
                            if dal is not None and fi not in pre_da:
                                sums, rows = pre[fi] if fi in pre else K.affine_act_bwd_reduce(dnodes[node], st.saved[fi].raw, None, None, 0)
                                K.plain_bwd_coeffs(sums, rows, None, st.saved[fi].raw.B, st.saved[fi].raw.C, dev, dal.data_ptr() + 4 * col, want_A=False)
                            wps[seg.weight.is_max] = P._wptr(arow, col)
                        target, acc = tgt(ia)
                        K.pool2_bwd_both(dnodes[node], st.saved[fa].ws.x, target, acc, wps[False], wps[True])
                        skip.add(ui + 1)
                        continue
                if len(unit) >= 3 or unit in prep:
                    # targets are claimed in reverse term order, like the unpaired reverse walk
                    terms = []
                    for fi in reversed(unit):
                        _, idx, seg, col, amat, row = flat[fi]
                        arow, dal = alpha_of(amat, row)
                        target, acc = tgt(idx)
                        terms.append((seg, st.saved[fi], (True, target, acc), (arow, col, dal)))
                    terms.reverse()
                    for (seg, _, _, _), (_, gl) in zip(terms, P.group_backward(terms, dnodes[node], prep.get(unit))):
                        put(seg, gl)
                    continue
                if len(unit) == 2:
                    fa, fb = unit
                    _, ia, sega, cola, amata, rowa = flat[fa]
                    _, ib, segb, colb, amatb, rowb = flat[fb]
                    sa, sb = st.saved[fa], st.saved[fb]
                    (arowa, dala), (arowb, dalb) = alpha_of(amata, rowa), alpha_of(amatb, rowb)
                    if sa.kind == "gn" and sb.kind == "gn":
                        tb, ab = tgt(ib)     # the later term first, like the unpaired reverse walk
                        ta, aa = tgt(ia)
                        (_, ga), (_, gb) = P.pair_backward(sega, sa, segb, sb, dnodes[node], (True, ta, aa), (True, tb, ab), None,
                                                          (arowa, cola, dala), (arowb, colb, dalb))
                        put(segb, gb)
                        put(sega, ga)
                        continue
                    todo = ((ib, segb, sb, colb, arowb, dalb), (ia, sega, sa, cola, arowa, dala))
                else:
                    fi, = unit
                    _, idx, seg, col, amat, row = flat[fi]
                    arow, dal = alpha_of(amat, row)
                    todo = ((idx, seg, st.saved[fi], col, arow, dal),)
                for idx, seg, s, col, arow, dal in todo:
                    target, acc = tgt(idx)
                    _, gl = P.seg_backward(seg, s, dnodes[node], True, target, acc, arow, col, dal,
                                           pre.get(unit[0]) if len(unit) == 1 else None, pre_se.get(unit[0]) if len(unit) == 1 else None,
                                           len(unit) == 1 and unit[0] in pre_da)
                    put(seg, gl)

        sb = SIDE_BWD if (SIDE_BWD is not None and batch_reduce) else None
        if sb is None:
            for node in reversed(range(nn)):
                run_node(node, all_units[node])
                if NODE_DONE_HOOK is not None:
                    NODE_DONE_HOOK()
        else:
            # two streams (train.SideSchedule): the terms that read a NODE stay on the main stream -- their input gradients are what the
            # next node's backward waits for -- and the terms that read a preprocess output go to the side stream, behind a flag
            # that says "the gradient of this node is complete".  The two sets accumulate into disjoint tensors (node gradients /
            # preprocess gradients, which only the side stream touches until the join below).
            split = _node_units_split(plan, sb.bwd_side_inputs)
            ready = sb.fork()
            side_tok = None
            for node in reversed(range(nn)):
                on_main, on_side = split[node]
                if on_side:
                    with sb.side(ready):
                        run_node(node, on_side)
                        side_tok = sb.side_signal()
                if on_main:
                    run_node(node, on_main)
                if NODE_DONE_HOOK is not None:
                    NODE_DONE_HOOK()
                if node > 0:
                    ready = sb.fork()     # every node-to-node edge out of node - 1 has run: its gradient is complete
            if side_tok is not None:
                sb.join(side_tok)
    for i in range(2):
        if not pre_started[i]:
            dpre[i].t.zero_()
    (t0, acc0), (t1, acc1) = dx_targets if dx_targets is not None else ((None, False), (None, False))
    if late_joins:
        SIDE_BWD.join_many(late_joins)
    if (SIDE_BWD is not None and plan.pairs and need_x0 and dx_targets is not None
            and (getattr(st, "pre0_side", False) or SIDE_PRE0_BWD)):
        # searched net: the gradient of a cell's FIRST input -- the skip of an up cell, the output of the cell before the previous
        # one for a down cell -- is not needed by the next cell of the backward walk (that one reads the gradient of the SECOND
        # input), so the backward of that preprocess op goes to the side stream; NetFn.backward joins its flag (st.side_tok) in
        # front of whatever touches that gradient buffer next
        sb = SIDE_BWD
        f = sb.fork()
        with sb.side(f):
            d0, g0 = P.seg_backward(plan.pre0, st.s_pre0, dpre[0], True, t0, acc0)
            st.side_tok = sb.side_signal()
        d1, g1 = P.seg_backward(plan.pre1, st.s_pre1, dpre[1], need_x1, t1, acc1)
    else:
        (d0, g0), (d1, g1) = P.pair_backward(plan.pre0, st.s_pre0, plan.pre1, st.s_pre1, dpre[0], (need_x0, t0, acc0), (need_x1, t1, acc1),
                                             dpre[1])
    put(plan.pre0, g0)
    put(plan.pre1, g1)
    return d0, d1, da1, da2, grads


class SearchedCellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x0, x1, *params):
        out, st = _run_forward(plan, x0, x1, None, None)
        ctx.plan, ctx.st = plan, st
        return out

    @staticmethod
    def backward(ctx, dout):
        dx0, dx1, _, _, grads = _run_backward(ctx.plan, ctx.st, dout, None, None, ctx.needs_input_grad[1],
                                              ctx.needs_input_grad[2], False)
        ctx.st = None
        return (None, dx0, dx1) + tuple(g if ctx.needs_input_grad[3 + i] else None for i, g in enumerate(grads))


class CellFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, plan, x0, x1, alpha1, alpha2, *params):
        a1 = alpha1.detach().contiguous()
        a2 = alpha2.detach().contiguous()
        out, st = _run_forward(plan, x0, x1, a1, a2)
        ctx.plan, ctx.st, ctx.a1, ctx.a2 = plan, st, a1, a2
        return out

    @staticmethod
    def backward(ctx, dout):
        want = ctx.needs_input_grad[3] or ctx.needs_input_grad[4]
        dx0, dx1, da1, da2, grads = _run_backward(ctx.plan, ctx.st, dout, ctx.a1, ctx.a2, ctx.needs_input_grad[1],
                                                  ctx.needs_input_grad[2], want)
        ctx.st = None
        return (None, dx0, dx1, da1 if ctx.needs_input_grad[3] else None, da2 if ctx.needs_input_grad[4] else None) + \
            tuple(g if ctx.needs_input_grad[5 + i] else None for i, g in enumerate(grads))


class MixedOpFn(torch.autograd.Function):
    """sum_k w_k * op_k(x) for one edge (cell.py:24-33) as a single autograd node."""

    @staticmethod
    def forward(ctx, segs, x, weights, *params):
        xv = K.as_view(x, "input")
        w = weights.detach().contiguous()
        out = None
        saved = []
        for k, seg in enumerate(segs):
            if out is None:
                shp = seg.weight.out_shape(xv)
                out = K.as_view(K.empty_ndhwc(*shp, xv.t.device))
            _, s = P.seg_forward(seg, xv, None, out, k > 0, w, k)
            saved.append(s)
        ctx.segs, ctx.saved, ctx.w, ctx.xv = segs, saved, w, xv
        plist = []
        for seg in segs:
            plist += seg.params()
        ctx.index = {id(p): i for i, p in enumerate(plist)}
        ctx.nparams = len(plist)
        return out.t

    @staticmethod
    def backward(ctx, dout):
        dv = K.as_view(dout, "grad_output")
        xv = ctx.xv
        need_dx = ctx.needs_input_grad[1]
        dx = K.like(xv) if need_dx else None
        dalpha = torch.zeros_like(ctx.w) if ctx.needs_input_grad[2] else None
        grads = [None] * ctx.nparams
        for k, (seg, s) in enumerate(zip(ctx.segs, ctx.saved)):
            _, gl = P.seg_backward(seg, s, dv, need_dx, dx, k > 0, ctx.w, k, dalpha)
            for p, g in zip(seg.params(), gl):
                if g is not None and getattr(p, "_n3d_grad", None) is None:
                    grads[ctx.index[id(p)]] = g
        ctx.saved = None
        return (None, dx.t if need_dx else None, dalpha) + tuple(g if ctx.needs_input_grad[3 + i] else None
                                                                   for i, g in enumerate(grads))


# =================================================================================================
# whole-net autograd node
# =================================================================================================
BF16_MAX_NODE_WIDTH = 8  # bf16 configuration: cells with at most this many channels per node store their activations in bf16
WHOLE_NET = True  # nets run stems + cells as ONE autograd node (NetFn); False: one node per stem / cell
# data-parallel trainers: called as CELL_DONE_HOOK(k) inside NetFn.backward when every parameter gradient of cell k (k = index
# into down_cells + up_cells; -1 = the stems, i.e. the end) has been launched -- the point where a gradient bucket can be handed
# to the all-reduce while the backward of the remaining cells goes on (train.Trainer)
CELL_DONE_HOOK = None
NODE_DONE_HOOK = None   # called (force=False) inside a cell's backward after every node and (force=True) after each stem: the
                        # side-stream schedule's finer cut points


class _NetPlan:
    """Stems and cells of a U-shaped net in execution order, with the wiring of unet.route as index triples."""

    def __init__(self, net, supernet):
        self.supernet = supernet
        self.stem0, self.stem1 = _single_segment(net.stem0), _single_segment(net.stem1)
        make = supernet_plan if supernet else searched_plan
        cells = list(net.down_cells) + list(net.up_cells)
        self.version = P.PLAN_VERSION[0]
        for c in cells:
            if c._plan is None or c._plan.version != P.PLAN_VERSION[0]:
                c._plan = make(c)
        self.cells = [c._plan for c in cells]
        # storage policy (BASELINE configs[4]): with net._n3d_storage == "bf16" the stems and the cells of the HBM-bound levels
        # (node width <= 8: the 128^3 .. 32^3 tensors, ~94 % of a step's bytes) keep their activations and activation
        # gradients in bf16; the deep cells (16 .. 64 channels on <= 32^3 voxels, latency-bound, L2-resident) stay fp32
        bf = getattr(net, "_n3d_storage", "fp32") == "bf16"
        for pl in self.cells:
            # (round 5: every primitive of the registry has bf16-storage kernels at these widths -- depthwise, pooling, SE gate, identity
            # next to the convs -- so the policy no longer depends on the genotype; the supernet's N-term kernels stay fp32)
            pl.dt = torch.bfloat16 if (bf and pl.c_node <= BF16_MAX_NODE_WIDTH and not supernet) else torch.float32
            # (round 6) the deep cells of the bf16 configuration: fp32 storage, bf16 operands in their MFMA conv kernels (N3D_MM_BF16)
            pl.mm_bf16 = bool(bf and not supernet and pl.dt == torch.float32)
        self.stem_dt = torch.bfloat16 if bf else torch.float32
        self.n_down = len(net.down_cells)
        # activations: 0 = stem0, 1 = stem1, 2 + k = cell k.  wiring[k] = (x0 index, x1 index, output index)
        self.wiring = []
        older, newer, kept = 0, 1, [0, 1]
        for k in range(self.n_down):
            self.wiring.append((older, newer, 2 + k))
            older, newer = newer, 2 + k
            kept.append(newer)
        kept.pop()
        for k in range(self.n_down, len(cells)):
            self.wiring.append((kept.pop(), newer, 2 + k))
            newer = 2 + k
        # cells whose output may stay node-planar (PLANAR_INNER): read by exactly ONE op -- the x1 preprocess conv of a later cell, a plain
        # 1x1x1 stride-1 conv -- and by nothing else; value = that conv's output channels (the shapes are checked when the cell runs)
        self.planar_reader = {}
        if not supernet:
            for k in range(len(cells) - 1):
                readers = [(j, side) for j, (i0, i1, _) in enumerate(self.wiring) for side, idx in ((0, i0), (1, i1)) if idx == 2 + k]
                if len(readers) == 1 and readers[0][1] == 1 and self.cells[k].pairs:
                    pre = self.cells[readers[0][0]].pre1
                    w = pre.weight
                    if (isinstance(w, P.DenseConvW) and w.k == 1 and w.stride == 1 and not w.transposed and pre.dropout is None
                            and pre.se_gate is None and self.cells[readers[0][0]].pairs):
                        self.planar_reader[k] = int(w.m.weight.shape[0])
        self.params = self.stem0.params() + self.stem1.params()
        self.offsets = []
        for pl in self.cells:
            self.offsets.append(len(self.params))
            self.params = self.params + pl.params


def net_plan(net, supernet):
    return _NetPlan(net, supernet)


def current(plan):
    """is a cached cell / net plan still built from the ops' current launch programs?"""
    return plan is not None and plan.version == P.PLAN_VERSION[0]


class NetFn(torch.autograd.Function):
    """Stems + every cell of a SearchedNet / KernelNet as one autograd node (nas.py:54-77, searched.py:93-110).

    With one node per cell, autograd sums the gradients of every activation that feeds two or three consumers (each cell
    output is the next cell's x1, the one after's x0 and possibly a skip) with separate add kernels.  Here the backward walk
    owns the gradient buffers: the first consumer to run writes, later ones accumulate in their data-gradient kernels."""

    @staticmethod
    def forward(ctx, nplan, x, a1d, a1u, a2d, a2u, *params):
        xv = K.as_view(x, "input")
        al = [a.detach().contiguous() if a is not None else None for a in (a1d, a1u, a2d, a2u)]
        sf = SIDE_FWD
        early = {}       # up cell k -> (View, saved state, flag id) of its skip-side preprocess op, launched on the side stream

        def spawn_pre0(act_index):
            """searched net, SIDE_FWD: the activation `act_index` exists now -- the up cell that takes it as its skip input does
            not need it before the whole lower part of the U has run, so its 1x1x1 preprocess op starts on the side stream here"""
            if sf is None or nplan.supernet:
                return
            # (the down cells' first preprocess op stays on the chain: the two preprocess ops of a down cell share their conv and epilogue
            # launches there, and splitting that pair measured slower than the join it saves)
            for k in range(nplan.n_down, len(nplan.wiring)):
                pl = nplan.cells[k]
                if nplan.wiring[k][0] == act_index and pl.pairs and pl.pre0.dropout is None:
                    f = sf.fork()
                    with sf.side(f), K.storage(pl.dt, getattr(pl, "mm_bf16", False)):
                        p0, s_pre0 = P.seg_forward(pl.pre0, K.as_view(acts[act_index], "x0"))
                        early[k] = (p0, s_pre0, sf.side_signal())

        # the net input needs no gradient (the usual case): a stem may run without storing its raw conv output (P.recompute_ok)
        rc = not ctx.needs_input_grad[1]
        if sf is not None:
            # the two stems are independent chains (conv, coefficients, epilogue): one of them on the side stream
            f = sf.fork()
            with sf.side(f), K.storage(nplan.stem_dt):
                s1, st1 = P.seg_forward(nplan.stem1, xv, recompute=rc)
                tok1 = sf.side_signal()
            with K.storage(nplan.stem_dt):
                s0, st0 = P.seg_forward(nplan.stem0, xv, recompute=rc)
            sf.join(tok1)
        else:
            with K.storage(nplan.stem_dt):
                s0, st0 = P.seg_forward(nplan.stem0, xv, recompute=rc)
                s1, st1 = P.seg_forward(nplan.stem1, xv, recompute=rc)
        acts, states = [s0.t, s1.t], []
        spawn_pre0(0)
        spawn_pre0(1)
        for k, (i0, i1, _) in enumerate(nplan.wiring):
            a1, a2 = (al[0], al[2]) if k < nplan.n_down else (al[1], al[3])
            out, st = _run_forward(nplan.cells[k], acts[i0], acts[i1], a1, a2, early.pop(k, None),
                                   planar=(PLANAR_OUT and k == len(nplan.wiring) - 1) or (PLANAR_INNER and nplan.planar_reader.get(k, 0)))
            acts.append(out)
            states.append(st)
            if k < nplan.n_down:
                spawn_pre0(2 + k)
        ctx.nplan, ctx.al, ctx.st0, ctx.st1, ctx.states = nplan, al, st0, st1, states
        return acts[-1]

    @staticmethod
    def backward(ctx, dout):
        nplan, al = ctx.nplan, ctx.al
        want_da = any(ctx.needs_input_grad[2:6])
        n_acts = 2 + len(nplan.cells)
        gbuf = [None] * n_acts
        gbuf[-1] = dout
        grads = [None] * len(nplan.params)
        das = [None] * 4
        pending = {}     # activation index -> flag id: its gradient buffer is being written by the side stream (SIDE_BWD)

        def settle(*idx):
            for i in idx:
                if i in pending:
                    SIDE_BWD.join(pending.pop(i))

        for k in reversed(range(len(nplan.cells))):
            i0, i1, io = nplan.wiring[k]
            down = k < nplan.n_down
            a1, a2 = (al[0], al[2]) if down else (al[1], al[3])
            settle(io)                                                      # read from the first launch on
            late = [pending.pop(i) for i in (i0, i1) if i in pending]       # only written, by the preprocess backward at the end
            targets = tuple((K.as_act(gbuf[i], "grad") if gbuf[i] is not None else None, gbuf[i] is not None) for i in (i0, i1))
            d0, d1, da1, da2, gl = _run_backward(nplan.cells[k], ctx.states[k], gbuf[io], a1, a2, True, True, want_da, targets,
                                                 own_dout=gbuf[io] is not dout, late_joins=late)
            tok = getattr(ctx.states[k], "side_tok", None)
            if tok is not None:
                pending[i0] = tok
            ctx.states[k] = None
            gbuf[io] = None
            gbuf[i0], gbuf[i1] = d0, d1
            for j, g in enumerate(gl):
                grads[nplan.offsets[k] + j] = g
            for slot, d in ((0 if down else 1, da1), (2 if down else 3, da2)):
                if d is not None:
                    das[slot] = d if das[slot] is None else das[slot].add_(d)
            if CELL_DONE_HOOK is not None:
                CELL_DONE_HOOK(k)
        settle(0, 1)
        need_x = ctx.needs_input_grad[1]
        dx = None
        n0 = len(nplan.stem0.params())
        for seg, st, g, off in ((nplan.stem1, ctx.st1, gbuf[1], n0), (nplan.stem0, ctx.st0, gbuf[0], 0)):
            if seg is nplan.stem1 and SIDE_BWD is not None and SIDE_STEM1_BWD and not need_x and not nplan.supernet:
                # the two stems' backwards are independent and produce parameter gradients only: one of them on the side stream,
                # whose 'done' flag the tail waits for anyway (no join of its own)
                f = SIDE_BWD.fork()
                with SIDE_BWD.side(f):
                    d, gl = P.seg_backward(seg, st, K.as_view(g, "grad"), need_x)
                    SIDE_BWD.side_signal()
            else:
                d, gl = P.seg_backward(seg, st, K.as_view(g, "grad"), need_x)
            for j, (p, gg) in enumerate(zip(seg.params(), gl)):
                if gg is not None and getattr(p, "_n3d_grad", None) is None:
                    grads[off + j] = gg
            if need_x:
                dx = d if dx is None else dx + d
            if NODE_DONE_HOOK is not None:
                NODE_DONE_HOOK(True)     # side-stream schedule: stem1's weight gradient starts under stem0's epilogue backward
        ctx.st0 = ctx.st1 = ctx.states = None
        if CELL_DONE_HOOK is not None:
            CELL_DONE_HOOK(-1)
        return (None, dx) + tuple(d if ctx.needs_input_grad[2 + i] else None for i, d in enumerate(das)) + \
            tuple(g if ctx.needs_input_grad[6 + i] else None for i, g in enumerate(grads))
