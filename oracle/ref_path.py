"""CPU oracle: a table-driven, functional restatement of the reference hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  The reference
implements the path as ``torch.nn`` module classes; the arithmetic itself lives
in third-party PyTorch (README pin torch==1.2.0; here torch 2.10 CPU).  This
file restates the same algorithm as pure functions over a flat
``{state_dict_name: tensor}`` dictionary, using ``torch.nn.functional`` on
whatever dtype the dictionary holds (fp32 = the reference's arithmetic,
fp64 = a tighter oracle).  Gradients come from autograd.

Reference lines followed (relative to /root/reference):
  primitive table            prim_ops.py:5-45
  op sequencing / GroupNorm  prim_ops.py:48-83   (G = 1 if C%16 else C//16)
  conv variants / padding    prim_ops.py:85-117  (pad = max(0, ceil((d(k-1)-s+1)/2)))
  SE gate                    prim_ops.py:119-153
  pooling / identity         prim_ops.py:155-174
  mixed op, cell DAG         cell.py:8-33, 35-82
  supernet U shape, alphas   nas.py:13-78, 81-135
  searched cell / net        searched.py:10-51, 54-111
  Dice loss                  loss.py:6-14
  genotype decoding          genotype.py:19-45
  train / search step order  train.py:117-128, search.py:211-238
"""
from __future__ import annotations

import math
import zlib
from collections import namedtuple

import numpy as np
import torch
import torch.nn.functional as F

__all__ = [
    "PRIMS", "DOWN_NAMES", "UP_NAMES", "NORM_NAMES", "Genotype", "G_CONV", "G_ALL",
    "NetCfg", "DEFAULT_CFG", "group_count", "conv_pad",
    "prim_param_specs", "convops_param_specs", "searched_param_specs", "supernet_param_specs",
    "fill_value", "make_params", "prim_forward", "convops_forward", "mixed_forward",
    "cell_forward", "searched_cell_forward", "searched_forward", "supernet_forward",
    "dice_loss", "parse_genotype", "supernet_genotype", "synthetic_patch",
    "adam_reference_steps",
]

# --------------------------------------------------------------------------------------
# primitive table (prim_ops.py:5-21).  kind: 'gn' identity-with-norm, 'se', 'conv', 'dw', 'pool'
# fields: kind, stride, dilation, transposed, pool_type
# --------------------------------------------------------------------------------------
PRIMS = {
    "identity":      ("gn",   1, 1, False, None),
    "se_conv":       ("se",   1, 1, False, None),
    "dil_conv":      ("conv", 1, 2, False, None),
    "dep_conv":      ("dw",   1, 1, False, None),
    "conv":          ("conv", 1, 1, False, None),
    "avg_pool":      ("pool", 2, 1, False, "avg"),
    "max_pool":      ("pool", 2, 1, False, "max"),
    "down_se_conv":  ("se",   2, 1, False, None),
    "down_dil_conv": ("conv", 2, 2, False, None),
    "down_dep_conv": ("dw",   2, 1, False, None),
    "down_conv":     ("conv", 2, 1, False, None),
    "up_se_conv":    ("se",   2, 1, True,  None),
    "up_dep_conv":   ("dw",   2, 1, True,  None),
    "up_conv":       ("conv", 2, 1, True,  None),
    "up_dil_conv":   ("conv", 2, 2, True,  None),
}
# ordered name lists: the order is the alpha-column order (prim_ops.py:23-45, cell.py:16-22)
DOWN_NAMES = ["avg_pool", "max_pool", "down_se_conv", "down_dil_conv", "down_dep_conv", "down_conv"]
UP_NAMES = ["up_se_conv", "up_dep_conv", "up_conv", "up_dil_conv"]
NORM_NAMES = ["identity", "se_conv", "dil_conv", "dep_conv", "conv"]

Genotype = namedtuple("Genotype", ["down", "up"])  # genotype.py:6

# benchmark genotypes chosen by SURVEY.md appendix D (the reference ships none)
G_CONV = Genotype(
    down=[("down_conv", 0), ("down_dil_conv", 1), ("down_conv", 1), ("conv", 2), ("dil_conv", 2), ("conv", 3)],
    up=[("conv", 0), ("up_conv", 1), ("up_conv", 1), ("dil_conv", 2), ("conv", 3), ("up_dil_conv", 1)],
)
G_ALL = Genotype(
    down=[("down_se_conv", 0), ("max_pool", 1), ("down_dep_conv", 0), ("se_conv", 2), ("avg_pool", 1), ("dep_conv", 3)],
    up=[("identity", 0), ("up_se_conv", 1), ("up_dep_conv", 1), ("se_conv", 2), ("up_dil_conv", 1), ("dep_conv", 3)],
)

NetCfg = namedtuple("NetCfg", ["in_channels", "init_n_kernels", "out_channels", "depth", "n_nodes", "channel_change"])
DEFAULT_CFG = NetCfg(4, 4, 3, 4, 3, True)  # config.yml:3-7,19-22,44-45,50,55


def group_count(c: int) -> int:
    """prim_ops.py:57"""
    return 1 if c % 16 != 0 else c // 16


def conv_pad(k: int, stride: int, dilation: int) -> int:
    """prim_ops.py:91"""
    return max(0, math.ceil((dilation * (k - 1) - stride + 1) / 2))


# --------------------------------------------------------------------------------------
# parameter inventories (names/shapes equal the reference modules' state_dict)
# --------------------------------------------------------------------------------------
def convops_param_specs(prefix, cin, cout, k=3, transposed=False, depthwised=False, norm=True):
    """ConvOps attribute names and torch-native shapes (prim_ops.py:93-110)."""
    out = []
    if depthwised:
        out += [(prefix + "depth_conv.weight", (cin, 1, k, k, k)), (prefix + "depth_conv.bias", (cin,)),
                (prefix + "point_conv.weight", (cout, cin, 1, 1, 1)), (prefix + "point_conv.bias", (cout,))]
    elif transposed:
        out += [(prefix + "conv.weight", (cin, cout, k, k, k)), (prefix + "conv.bias", (cout,))]
    else:
        out += [(prefix + "conv.weight", (cout, cin, k, k, k)), (prefix + "conv.bias", (cout,))]
    if norm:
        out += [(prefix + "norm.weight", (cout,)), (prefix + "norm.bias", (cout,))]
    return out


def prim_param_specs(prefix, name, c):
    kind, stride, dil, transposed, _ = PRIMS[name]
    if kind == "gn":
        return [(prefix + "norm.weight", (c,)), (prefix + "norm.bias", (c,))]
    if kind == "pool":
        return []
    if kind == "se":
        out = [(prefix + "fc.0.weight", (1, c)), (prefix + "fc.0.bias", (1,)),
               (prefix + "fc.2.weight", (c, 1)), (prefix + "fc.2.bias", (c,))]
        if stride > 1:
            out += [(prefix + "conv.weight", (c, c, 3, 3, 3)), (prefix + "conv.bias", (c,)),
                    (prefix + "norm.weight", (c,)), (prefix + "norm.bias", (c,))]
        # module registration order in the reference: norm (BaseOp) first, then fc, conv
        return out
    return convops_param_specs(prefix, c, c, 3, transposed, kind == "dw", norm=True)


def _u_shape(cfg: NetCfg):
    """Channel bookkeeping of the U shape (nas.py:25-49 / searched.py:68-90).

    Returns (c_stem, down[(c0,c1,c_node)], up[(c0,c1,c_node)], c_head_in)."""
    c0 = c1 = cfg.n_nodes * cfg.init_n_kernels
    c_node = cfg.init_n_kernels
    c_stem = c0
    skips = [c0, c1]
    down = []
    for _ in range(cfg.depth):
        c_node = 2 * c_node if cfg.channel_change else c_node
        down.append((c0, c1, c_node))
        c0, c1 = c1, cfg.n_nodes * c_node
        skips.append(c1)
    skips.pop()
    up = []
    for _ in range(cfg.depth + 1):
        c0 = skips.pop()
        up.append((c0, c1, c_node))
        c1 = cfg.n_nodes * c_node
        c_node = c_node // 2 if cfg.channel_change else c_node
    return c_stem, down, up, c1


def _cell_edges(n_nodes, downward):
    """(node, input_index, stride) per MixedOp in cell order (cell.py:54-59)."""
    edges = []
    for node in range(n_nodes):
        for i in range(node + 2):
            if downward:
                edges.append((node, i, 2 if i <= 1 else 1))
            else:
                edges.append((node, i, 2 if i == 1 else 1))
    return edges


def _edge_prims(stride, downward):
    if stride == 1:
        return NORM_NAMES
    return DOWN_NAMES if downward else UP_NAMES


def searched_param_specs(cfg: NetCfg, gene: Genotype):
    c_stem, down, up, c_head = _u_shape(cfg)
    specs = convops_param_specs("stem0.", cfg.in_channels, c_stem, 1)
    specs += convops_param_specs("stem1.", cfg.in_channels, c_stem, 3)
    for tag, cells, genolist in (("down_cells", down, gene.down), ("up_cells", up, gene.up)):
        for ci, (c0, c1, cn) in enumerate(cells):
            p = "%s.%d." % (tag, ci)
            specs += convops_param_specs(p + "preprocess0.", c0, cn, 1)
            specs += convops_param_specs(p + "preprocess1.", c1, cn, 1)
            for oi, (name, _) in enumerate(genolist):
                specs += prim_param_specs(p + "_ops.%d." % oi, name, cn)
    specs += convops_param_specs("last_conv.0.", c_head, cfg.out_channels, 1, norm=False)
    return specs


def supernet_param_specs(cfg: NetCfg, normal_w_share=False):
    n_edges = sum(range(2, 2 + cfg.n_nodes))
    specs = [("alpha2_down", (n_edges, len(DOWN_NAMES))), ("alpha2_up", (n_edges, len(UP_NAMES))),
             ("alpha1_down", (n_edges, len(NORM_NAMES)))]
    if not normal_w_share:
        specs.append(("alpha1_up", (n_edges, len(NORM_NAMES))))
    c_stem, down, up, c_head = _u_shape(cfg)
    k = "kernel."
    specs += convops_param_specs(k + "stem0.", cfg.in_channels, c_stem, 1)
    specs += convops_param_specs(k + "stem1.", cfg.in_channels, c_stem, 3)
    for tag, cells, downward in (("down_cells", down, True), ("up_cells", up, False)):
        for ci, (c0, c1, cn) in enumerate(cells):
            p = "%s%s.%d." % (k, tag, ci)
            specs += convops_param_specs(p + "preprocess0.", c0, cn, 1)
            specs += convops_param_specs(p + "preprocess1.", c1, cn, 1)
            for ei, (_, _, stride) in enumerate(_cell_edges(cfg.n_nodes, downward)):
                for pi, name in enumerate(_edge_prims(stride, downward)):
                    specs += prim_param_specs(p + "_ops.%d._ops.%d." % (ei, pi), name, cn)
    specs += convops_param_specs(k + "last_conv.0.", c_head, cfg.out_channels, 1, norm=False)
    return specs


# --------------------------------------------------------------------------------------
# closed-form parameter fill keyed by state-dict name; the same function is
# applied to the reference modules (fixture generation), to this oracle and to the HIP modules.
# --------------------------------------------------------------------------------------
def fill_value(name: str, shape, salt: int = 0) -> np.ndarray:
    rng = np.random.default_rng([zlib.crc32(name.encode()), salt])
    n = rng.standard_normal(tuple(shape))
    leaf = name.rsplit(".", 1)[-1]
    if "alpha" in name:
        v = 0.5 * n
    elif "norm." in name:
        v = 1.0 + 0.2 * n if leaf == "weight" else 0.1 * n
    elif leaf == "bias":
        v = 0.1 * n
    else:
        fan = max(1, int(np.prod(shape[1:]))) if len(shape) > 1 else 1
        v = n * (1.0 / math.sqrt(fan))
    return v.astype(np.float64)


def make_params(specs, dtype=torch.float32, salt=0, requires_grad=False):
    out = {}
    for name, shape in specs:
        t = torch.from_numpy(fill_value(name, shape, salt)).to(dtype)
        out[name] = t.requires_grad_(requires_grad)
    return out


# --------------------------------------------------------------------------------------
# forward restatement
# --------------------------------------------------------------------------------------
def _gn(P, prefix, x):
    c = x.shape[1]
    return F.group_norm(x, group_count(c), P[prefix + "norm.weight"], P[prefix + "norm.bias"], 1e-5)


def _weight_call(P, prefix, x, k, stride, dil, transposed, depthwised):
    pad = conv_pad(k, stride, dil)
    opad = 0 if stride == 1 else 1
    if depthwised:
        c = x.shape[1]
        # NB: the depthwise variants never receive `dilation` (prim_ops.py:95-97,105-106)
        if transposed:
            x = F.conv_transpose3d(x, P[prefix + "depth_conv.weight"], P[prefix + "depth_conv.bias"],
                                   stride=stride, padding=pad, output_padding=opad, groups=c)
        else:
            x = F.conv3d(x, P[prefix + "depth_conv.weight"], P[prefix + "depth_conv.bias"],
                         stride=stride, padding=pad, groups=c)
        return F.conv3d(x, P[prefix + "point_conv.weight"], P[prefix + "point_conv.bias"])
    if transposed:
        return F.conv_transpose3d(x, P[prefix + "conv.weight"], P[prefix + "conv.bias"], stride=stride,
                                  padding=pad, output_padding=opad, dilation=dil)
    return F.conv3d(x, P[prefix + "conv.weight"], P[prefix + "conv.bias"], stride=stride, padding=pad,
                    dilation=dil)


def convops_forward(P, prefix, x, k=3, stride=1, dil=1, transposed=False, depthwised=False,
                    order="weight_norm_act", drop_mask=None):
    """BaseOp.forward + ConvOps.weight_call (prim_ops.py:68-83,111-117).

    ``drop_mask``: optional (B, Cin) tensor already scaled by 1/(1-p); stands in for the
    Dropout3d that the reference applies *before* the weight op (prim_ops.py:72-73)."""
    for tok in order.split("_"):
        if tok == "weight":
            if drop_mask is not None:
                x = x * drop_mask[:, :, None, None, None]
            x = _weight_call(P, prefix, x, k, stride, dil, transposed, depthwised)
        elif tok == "norm":
            if (prefix + "norm.weight") in P:
                x = _gn(P, prefix, x)
        elif tok == "act":
            x = F.relu(x)
        else:
            raise Warning("Unrecognized op: %s" % tok)
    return x


def _se_gate(P, prefix, x):
    """prim_ops.py:133-139,148-151: mean over DHW -> Linear(C,1) -> ReLU -> Linear(1,C) -> sigmoid."""
    m = x.mean(dim=(2, 3, 4))
    h = F.relu(F.linear(m, P[prefix + "fc.0.weight"], P[prefix + "fc.0.bias"]))
    g = torch.sigmoid(F.linear(h, P[prefix + "fc.2.weight"], P[prefix + "fc.2.bias"]))
    return g[:, :, None, None, None]


def prim_forward(P, prefix, name, x):
    """One registry primitive OPS[name](C) applied to x (prim_ops.py:5-21 + classes)."""
    kind, stride, dil, transposed, pool = PRIMS[name]
    if kind == "gn":  # IdentityOp is GroupNorm -> ReLU, not a pure identity (prim_ops.py:170-174)
        return F.relu(_gn(P, prefix, x))
    if kind == "pool":  # ops_order='weight': no norm/act (prim_ops.py:157)
        return F.avg_pool3d(x, 2, 2) if pool == "avg" else F.max_pool3d(x, 2, 2)
    if kind == "se":
        xs = x * _se_gate(P, prefix, x)
        if stride == 1:  # scale only: ops_order forced to 'weight' (prim_ops.py:127-128,152)
            return xs
        pad = conv_pad(3, stride, 1)
        if transposed:
            y = F.conv_transpose3d(xs, P[prefix + "conv.weight"], P[prefix + "conv.bias"], stride=stride,
                                   padding=pad, output_padding=1)
        else:
            y = F.conv3d(xs, P[prefix + "conv.weight"], P[prefix + "conv.bias"], stride=stride, padding=pad)
        return _gn(P, prefix, y)  # 'weight_norm': no ReLU (prim_ops.py:126)
    return convops_forward(P, prefix, x, 3, stride, dil, transposed, kind == "dw", "weight_norm_act")


def mixed_forward(P, prefix, x, weights, stride, downward):
    """MixedOp.forward (cell.py:24-33): left-to-right sum of w_k * op_k(x)."""
    names = _edge_prims(stride, downward)
    acc = 0
    for k, name in enumerate(names):
        acc = acc + weights[k] * prim_forward(P, prefix + "_ops.%d." % k, name, x)
    return acc


def _cell_pre(P, prefix, x0, x1, downward):
    x0 = convops_forward(P, prefix + "preprocess0.", x0, 1, 2 if downward else 1, order="act_weight_norm")
    x1 = convops_forward(P, prefix + "preprocess1.", x1, 1, 1, order="act_weight_norm")
    return x0, x1


def cell_forward(P, prefix, x0, x1, alpha1, alpha2, n_nodes, downward):
    """Cell.forward (cell.py:66-82).  The global edge counter indexes both alpha matrices."""
    x0, x1 = _cell_pre(P, prefix, x0, x1, downward)
    xs = [x0, x1]
    e = 0
    for node in range(n_nodes):
        acc = 0
        for i, x in enumerate(list(xs)):
            stride = (2 if i <= 1 else 1) if downward else (2 if i == 1 else 1)
            w = alpha1[e] if stride == 1 else alpha2[e]
            acc = acc + mixed_forward(P, prefix + "_ops.%d." % e, x, w, stride, downward)
            e += 1
        xs.append(acc)
    return torch.cat(xs[-n_nodes:], dim=1)


def searched_cell_forward(P, prefix, x0, x1, genolist, n_nodes, downward):
    """SearchedCell.forward (searched.py:37-51)."""
    x0, x1 = _cell_pre(P, prefix, x0, x1, downward)
    xs = [x0, x1]
    for node in range(n_nodes):
        (na, ia), (nb, ib) = genolist[2 * node], genolist[2 * node + 1]
        a = prim_forward(P, prefix + "_ops.%d." % (2 * node), na, xs[ia])
        b = prim_forward(P, prefix + "_ops.%d." % (2 * node + 1), nb, xs[ib])
        xs.append(0 + a + b)
    return torch.cat(xs[-n_nodes:], dim=1)


def _u_forward(P, k, x, cfg, cell_fn, drop_mask, return_logits):
    s0 = convops_forward(P, k + "stem0.", x, 1, 1, order="weight_norm")
    s1 = convops_forward(P, k + "stem1.", x, 3, 2, order="weight_norm")
    stack = [s0, s1]
    for ci in range(cfg.depth):
        s0, s1 = s1, cell_fn("%sdown_cells.%d." % (k, ci), s0, s1, True)
        stack.append(s1)
    stack.pop()
    for ci in range(cfg.depth + 1):
        s0 = stack.pop()
        s1 = cell_fn("%sup_cells.%d." % (k, ci), s0, s1, False)
    logits = convops_forward(P, k + "last_conv.0.", s1, 1, 1, order="weight", drop_mask=drop_mask)
    probs = torch.sigmoid(logits)
    return (probs, logits) if return_logits else probs


def searched_forward(P, x, gene: Genotype, cfg: NetCfg = DEFAULT_CFG, drop_mask=None, return_logits=False):
    """SearchedNet.forward (searched.py:95-111); dropout off unless a mask is supplied."""
    def cell_fn(prefix, a, b, downward):
        return searched_cell_forward(P, prefix, a, b, gene.down if downward else gene.up, cfg.n_nodes, downward)
    return _u_forward(P, "", x, cfg, cell_fn, drop_mask, return_logits)


def supernet_forward(P, x, cfg: NetCfg = DEFAULT_CFG, drop_mask=None, return_logits=False, normal_w_share=False):
    """ShellNet.forward -> KernelNet.forward (nas.py:121-126, 54-78)."""
    a1d = F.softmax(P["alpha1_down"], dim=-1)
    a1u = a1d if normal_w_share else F.softmax(P["alpha1_up"], dim=-1)
    a2d = F.softmax(P["alpha2_down"], dim=-1)
    a2u = F.softmax(P["alpha2_up"], dim=-1)

    def cell_fn(prefix, a, b, downward):
        return cell_forward(P, prefix, a, b, a1d if downward else a1u, a2d if downward else a2u,
                            cfg.n_nodes, downward)
    return _u_forward(P, "kernel.", x, cfg, cell_fn, drop_mask, return_logits)


def dice_loss(p, t, smooth=1e-6):
    """WeightedDiceLoss.forward (loss.py:12-14)."""
    ax = (-1, -2, -3)
    return 1 - torch.mean((2 * torch.sum(p * t, dim=ax) + smooth) / (torch.sum(p, dim=ax) + torch.sum(t, dim=ax) + smooth))


# --------------------------------------------------------------------------------------
# genotype decoding (genotype.py:19-45)
# --------------------------------------------------------------------------------------
def parse_genotype(alpha1, alpha2, n_nodes, downward=True):
    alpha1 = np.asarray(alpha1)
    alpha2 = np.asarray(alpha2)
    res = []
    e = 0
    for n_in in range(2, 2 + n_nodes):
        cand = []
        for edge in range(n_in):
            strided = (edge < 2) if downward else (edge == 1)
            if strided:
                names = DOWN_NAMES if downward else UP_NAMES
                j = int(np.argmax(alpha2[e]))
                cand.append((alpha2[e][j] * len(names) / len(NORM_NAMES), names[j], edge))
            else:
                j = int(np.argmax(alpha1[e]))
                cand.append((alpha1[e][j], NORM_NAMES[j], edge))
            e += 1
        cand.sort()
        res += [(c[1], c[2]) for c in cand[-2:]]
    return res


def supernet_genotype(P, n_nodes, normal_w_share=False):
    """ShellNet.get_gene (nas.py:128-135)."""
    sm = lambda n: F.softmax(P[n].detach().float(), dim=-1).cpu().numpy()
    a1u = "alpha1_down" if normal_w_share else "alpha1_up"
    return Genotype(down=parse_genotype(sm("alpha1_down"), sm("alpha2_down"), n_nodes, True),
                    up=parse_genotype(sm(a1u), sm("alpha2_up"), n_nodes, False))


# --------------------------------------------------------------------------------------
# synthetic data and the optimiser pin (train.py:49,121-128)
# --------------------------------------------------------------------------------------
def synthetic_patch(batch, size, seed=1234, in_channels=4):
    """x: masked clipped Gaussian in [10,110] inside a centred ball, 0 outside; t: nested balls."""
    rng = np.random.default_rng(seed)
    g = np.arange(size, dtype=np.float64) - (size - 1) / 2.0
    r = np.sqrt(g[:, None, None] ** 2 + g[None, :, None] ** 2 + g[None, None, :] ** 2)
    mask = (r <= 0.45 * size)
    x = np.clip(50.0 + 25.0 * rng.standard_normal((batch, in_channels, size, size, size)), 10.0, 110.0) * mask
    t = np.stack([(r <= 0.22 * size), (r <= 0.30 * size), (r <= 0.12 * size)]).astype(np.float32)
    t = np.broadcast_to(t, (batch,) + t.shape).copy()
    return x.astype(np.float32), t


def adam_reference_steps(P, loss_fn, n_steps, lr=1e-3):
    """zero_grad -> forward -> loss -> backward -> Adam.step, n times (train.py:121-128, Adam defaults)."""
    params = [p for p in P.values() if p.requires_grad]
    opt = torch.optim.Adam(params, lr=lr)
    losses = []
    for _ in range(n_steps):
        opt.zero_grad()
        loss = loss_fn(P)
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
    return losses
