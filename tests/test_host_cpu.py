"""CPU-only tests of the host side: C-ABI symbol coverage, module surface / state-dict parity, segment
building, error behaviour, genotype decoding.  No kernel is launched here."""
import importlib
import os
import re
import sys

import numpy as np
import pytest
import torch

import golden_common as gc
from oracle import ref_path as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    from nas_3d_unet_amd import _lib
    hdr = open(os.path.join(ROOT, "include", "n3d.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(n3d_[a-zA-Z0-9_]+)\s*\(", hdr))
    declared -= {"n3d_conv_geom", "n3d_pack_job", "n3d_final_job"}
    assert declared, "no declarations parsed"
    lib = _lib.load()
    for name in sorted(declared):
        assert hasattr(lib, name), "libn3d.so does not export %s" % name
    assert declared == set(_lib.PROTOTYPES), (declared ^ set(_lib.PROTOTYPES))
    assert lib.n3d_version() >= 1
    assert isinstance(lib.n3d_last_error(), bytes)


def test_registry_surface(golden):
    from nas_3d_unet_amd import prim_ops
    g = golden("small")
    assert list(prim_ops.OPS.keys()) == list(g["registry/OPS"])
    assert prim_ops.DownOps == list(g["registry/DownOps"])
    assert prim_ops.UpOps == list(g["registry/UpOps"])
    assert prim_ops.NormOps == list(g["registry/NormOps"])


def test_state_dict_inventory_matches_reference(golden):
    from nas_3d_unet_amd import nas, searched
    g = golden("small")
    for gname in ("G_CONV", "G_ALL"):
        m = searched.SearchedNet(4, 4, 3, 4, 3, True, searched.Genotype(*getattr(orc, gname)))
        mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
        ref = dict(zip(g["inventory/searched/%s/names" % gname], g["inventory/searched/%s/shapes" % gname]))
        assert mine == ref
    m = nas.ShellNet(4, 4, 3, 4, 3, False, True)
    mine = {k: str(tuple(v.shape)) for k, v in m.state_dict().items()}
    assert mine == dict(zip(g["inventory/supernet/names"], g["inventory/supernet/shapes"]))
    assert [n for n, _ in m._alphas] == ["alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up"]


def test_segment_building_and_errors():
    from nas_3d_unet_amd import prim_ops
    from nas_3d_unet_amd._lib import N3DError
    op = prim_ops.ConvOps(4, 8, ops_order="weight_norm_act")
    segs = op._build_segments()
    assert len(segs) == 1 and segs[0].norm is op.norm and segs[0].relu_out and not segs[0].relu_in
    op = prim_ops.ConvOps(4, 8, kernel_size=1, ops_order="act_weight_norm")
    segs = op._build_segments()
    assert len(segs) == 1 and segs[0].relu_in and not segs[0].relu_out and segs[0].norm is op.norm
    op = prim_ops.ConvOps(4, 8, ops_order="norm_weight_act")       # not a canonical order: two segments
    assert len(op._build_segments()) == 2
    assert len(prim_ops.SEConvOp(8, 8)._build_segments()) == 1      # stride 1: forced to 'weight'
    assert prim_ops.SEConvOp(8, 8).norm is None and prim_ops.SEConvOp(8, 8, stride=2).norm is not None
    with pytest.raises(Warning):
        prim_ops.ConvOps(4, 4, ops_order="weight_bogus")(torch.zeros(1, 4, 2, 2, 2))
    with pytest.raises(NotImplementedError):
        prim_ops.PoolingOp(4, 4, "median")
    with pytest.raises(N3DError):                                    # no CPU fallback
        prim_ops.OPS["conv"](4)(torch.zeros(1, 4, 4, 4, 4))
    # group rule and padding formula of the reference
    assert [prim_ops.P.group_count(c) for c in (4, 8, 12, 16, 32, 64)] == [1, 1, 1, 1, 2, 4]
    assert [prim_ops._padding(3, s, d) for s, d in ((1, 1), (1, 2), (2, 1), (2, 2))] == [1, 2, 1, 2]
    assert prim_ops._padding(1, 1, 1) == 0 and prim_ops._padding(1, 2, 1) == 0


@pytest.mark.parametrize("key", gc.geno_cases())
def test_genotype_parser(golden, key):
    from nas_3d_unet_amd.genotype import GenoParser
    g = golden("small")
    p = GenoParser(3)
    a1 = gc.case_alpha_matrix(key + "/a1", 9, 5)
    gd = p.parse(a1, gc.case_alpha_matrix(key + "/a2d", 9, 6), True)
    gu = p.parse(a1, gc.case_alpha_matrix(key + "/a2u", 9, 4), False)
    assert [n for n, _ in gd] == list(g[key + "/down_names"]) and [i for _, i in gd] == list(g[key + "/down_idx"])
    assert [n for n, _ in gu] == list(g[key + "/up_names"]) and [i for _, i in gu] == list(g[key + "/up_idx"])


@pytest.mark.parametrize("key", ["tie/%s/%d" % (d, i) for d in ("down", "up") for i in range(4)])
def test_genotype_parser_on_near_tied_scores(golden, key):
    """index work is bit-exact: a stride-2 edge whose rescaled score ties (or is one float32 step off) a stride-1 edge's -- the
    decoded genotype then hangs on the reference's order of operations, (a * n) / 5 in float32 (genotype.py:35,38); every case was
    kept because a precomputed-ratio decoder decodes it differently (make_golden.gen_geno_ties)"""
    from nas_3d_unet_amd.genotype import GenoParser
    g = golden("geno_ties")
    got = GenoParser(3).parse(g[key + "/a1"], g[key + "/a2"], "/down/" in key)
    assert [n for n, _ in got] == list(g[key + "/names"]) and [i for _, i in got] == list(g[key + "/idx"])


def test_fresh_shellnet_genotype_is_the_degenerate_one():
    from nas_3d_unet_amd import nas
    gene = nas.ShellNet(4, 4, 3, 2, 3, False, True).get_gene()   # zero alphas
    assert [n for n, _ in gene.down][:3] == ["avg_pool", "avg_pool", "avg_pool"]
    assert gene.up[0][0] == "identity"


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="reference sources only exist in the build container")
def test_dropin_under_the_reference_network_files():
    """The reference's nas.py / searched.py, UNCHANGED, import this repo's prim_ops / cell:
    the nets build, and their state_dicts equal those built on the reference's own modules."""
    import nas_3d_unet_amd.cell, nas_3d_unet_amd.genotype, nas_3d_unet_amd.prim_ops
    saved = {k: sys.modules.get(k) for k in ("prim_ops", "cell", "genotype", "nas", "searched")}
    sys.path.insert(0, "/root/reference")
    try:
        for k in saved:
            sys.modules.pop(k, None)
        ref_nas = importlib.import_module("nas")
        ref_searched = importlib.import_module("searched")
        ref_sd = {k: tuple(v.shape) for k, v in ref_nas.ShellNet(4, 4, 3, 4, 3, False, True).state_dict().items()}
        ref_sd2 = {k: tuple(v.shape) for k, v in
                   ref_searched.SearchedNet(4, 4, 3, 4, 3, True, ref_searched.Genotype(*orc.G_ALL)).state_dict().items()}
        for k in saved:
            sys.modules.pop(k, None)
        sys.modules["prim_ops"] = nas_3d_unet_amd.prim_ops
        sys.modules["cell"] = nas_3d_unet_amd.cell
        sys.modules["genotype"] = nas_3d_unet_amd.genotype
        drop_nas = importlib.import_module("nas")
        drop_searched = importlib.import_module("searched")
        assert drop_nas.Cell is nas_3d_unet_amd.cell.Cell
        m = drop_nas.ShellNet(4, 4, 3, 4, 3, False, True)
        assert {k: tuple(v.shape) for k, v in m.state_dict().items()} == ref_sd
        m2 = drop_searched.SearchedNet(4, 4, 3, 4, 3, True, drop_searched.Genotype(*orc.G_ALL))
        assert {k: tuple(v.shape) for k, v in m2.state_dict().items()} == ref_sd2
        assert isinstance(m2.down_cells[0]._ops[0], nas_3d_unet_amd.prim_ops.SEConvOp)
        # round 6: the reference takes ANY init_n_kernels (nas.py:13-49, searched.py:55-90).  Over this repo's prim_ops / cell such a net
        # builds with the reference's parameter shapes; its ops and cells mark themselves for the zero-padded per-op path
        # (prim_ops._OpTwin; the forward itself needs the GPU: tests/test_gpu_dropin.py), and the twin of an op keeps the op's names
        m6 = drop_searched.SearchedNet(4, 6, 3, 2, 3, True, drop_searched.Genotype(*orc.G_ALL))
        for k in saved:
            sys.modules.pop(k, None)
        ref6 = importlib.import_module("searched").SearchedNet(4, 6, 3, 2, 3, True, importlib.import_module("searched").Genotype(*orc.G_ALL))
        assert {k: tuple(v.shape) for k, v in m6.state_dict().items()} == {k: tuple(v.shape) for k, v in ref6.state_dict().items()}
        assert m6.stem0._odd_channels() and m6.up_cells[-1].preprocess0._odd_channels() and not m6.last_conv[0]._odd_channels()
        assert not m6.down_cells[0].preprocess0._odd_channels()     # 18 -> 12 channels: a dense conv takes any input count as it is
        for op in [m6.stem0, m6.stem1] + list(m6.up_cells[-1]._ops):
            tw = nas_3d_unet_amd.prim_ops._OpTwin(op)
            tw.embed(op)
            tp = dict(tw.twin.named_parameters())
            for n, r in op.named_parameters():
                assert all(a % 4 == 0 or a == b or a in (1, 3) for a, b in zip(tp[n].shape, r.shape)), (n, tp[n].shape)
                lead = tp[n].detach()[tuple(slice(0, k) for k in r.shape)]
                assert torch.equal(lead, r.detach()) and float(tp[n].detach().abs().double().sum()) == float(r.detach().abs().double().sum())   # the rest zero
    finally:
        sys.path.remove("/root/reference")
        for k, v in saved.items():
            sys.modules.pop(k, None)
            if v is not None:
                sys.modules[k] = v


def test_plateau_lr_matches_torch_scheduler():
    """PlateauLR = host logic of ReduceLROnPlateau(optim, factor=0.5) (train.py:50,77; search.py:105-106,155-156)"""
    import numpy as np
    import torch
    from nas_3d_unet_amd.train import PlateauLR
    rng = np.random.default_rng(5)
    for trial in range(6):
        losses = np.abs(1.0 / (1 + 0.05 * np.arange(120)) + 0.02 * rng.standard_normal(120)) if trial % 2 else np.full(60, 0.5) + 1e-6 * rng.standard_normal(60)
        p = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([p])
        ref = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5)
        state = {"lr": 1e-3}
        mine = PlateauLR(lambda: state["lr"], lambda v: state.__setitem__("lr", v))
        for l in losses:
            ref.step(float(l))
            mine.step(float(l))
            assert abs(opt.param_groups[0]["lr"] - state["lr"]) < 1e-15
        assert state["lr"] < 1e-3  # the sequences are long enough to trigger at least one reduction


def test_checkpoint_interop_with_torch_adam_and_scheduler(tmp_path):
    """the trainer's flat Adam state and plateau scheduler round-trip through torch.optim.Adam /
    ReduceLROnPlateau state-dicts in the reference's checkpoint layout (train.py:85-93, 52-67)."""
    import torch
    from nas_3d_unet_amd import checkpoint as ck, searched
    from nas_3d_unet_amd.train import Trainer
    gene = searched.Genotype(down=[("down_conv", 0), ("down_dil_conv", 1), ("down_conv", 1), ("conv", 2), ("dil_conv", 2), ("conv", 3)],
                             up=[("conv", 0), ("up_conv", 1), ("up_conv", 1), ("dil_conv", 2), ("conv", 3), ("up_dil_conv", 1)])
    net = searched.SearchedNet(4, 4, 3, 2, 3, True, gene)
    tr = Trainer(net, graph=False)                       # construction only: no kernel runs on the CPU
    g = torch.Generator().manual_seed(0)
    tr.fp.exp_avg.copy_(torch.randn(tr.fp.numel, generator=g))
    tr.fp.exp_avg_sq.copy_(torch.rand(tr.fp.numel, generator=g))
    tr.fp.step.fill_(7)
    tr.set_lr(2.5e-4)
    tr.scheduler.best, tr.scheduler.num_bad_epochs, tr.scheduler.last_epoch = 0.31, 4, 12
    sd = ck.train_state_dicts(tr, epoch=12, history={"loss": [0.5]}, best_loss=0.31)
    assert set(sd) == {"epoch", "history", "model_param", "optim", "scheduler", "best_loss"}
    path = tmp_path / "last.pth"
    torch.save(sd, path)
    sd2 = torch.load(path, weights_only=False)
    # the reference side: plain torch objects accept the dictionaries
    ref_net = searched.SearchedNet(4, 4, 3, 2, 3, True, gene)
    ref_net.load_state_dict(sd2["model_param"])
    opt = torch.optim.Adam(ref_net.parameters())
    opt.load_state_dict(sd2["optim"])
    sch = torch.optim.lr_scheduler.ReduceLROnPlateau(opt, factor=0.5)
    sch.load_state_dict(sd2["scheduler"])
    assert opt.param_groups[0]["lr"] == 2.5e-4 and sch.best == 0.31 and sch.num_bad_epochs == 4
    p0 = list(ref_net.parameters())[3]
    o = tr.fp.offsets[3]
    assert torch.equal(opt.state[p0]["exp_avg"], tr.fp.exp_avg[o:o + p0.numel()].view(p0.shape))
    assert float(opt.state[p0]["step"]) == 7
    # and back: a checkpoint written by torch objects resumes the trainer
    tr2 = Trainer(searched.SearchedNet(4, 4, 3, 2, 3, True, gene), graph=False)
    back = {"epoch": 12, "history": {}, "model_param": ref_net.state_dict(), "optim": opt.state_dict(), "scheduler": sch.state_dict(), "best_loss": 0.31}
    epoch, _, best = ck.load_train_state_dicts(tr2, back)
    assert epoch == 13 and best == 0.31 and tr2.lr == 2.5e-4 and int(tr2.fp.step) == 7
    for p, o in zip(tr.fp.params, tr.fp.offsets):   # (the flat buffers pad every tensor to a multiple of 4 elements)
        n = p.numel()
        assert torch.equal(tr2.fp.exp_avg[o:o + n], tr.fp.exp_avg[o:o + n]) and torch.equal(tr2.fp.exp_avg_sq[o:o + n], tr.fp.exp_avg_sq[o:o + n])
    assert torch.equal(tr2.fp.flat, tr.fp.flat) and tr2.scheduler.num_bad_epochs == 4
    # round 6: a net with odd channel counts is trained as its zero-padded twin -- the checkpoint must still hold the
    # REFERENCE's shapes (moments cut back through the twin's index maps) and resume from a plain torch checkpoint of the same net
    pnet = searched.SearchedNet(4, 6, 3, 2, 3, True, gene)
    ptr = Trainer(pnet, graph=False)
    assert ptr._twin is not None
    ptr.fp.exp_avg.copy_(torch.randn(ptr.fp.numel, generator=g) * ptr._pad_mask)      # (padded entries: Adam never moves them)
    ptr.fp.exp_avg_sq.copy_(torch.rand(ptr.fp.numel, generator=g) * ptr._pad_mask)
    ptr.fp.step.fill_(5)
    psd = ck.train_state_dicts(ptr, epoch=3, history={}, best_loss=0.4)
    pref = searched.SearchedNet(4, 6, 3, 2, 3, True, gene)
    pref.load_state_dict(psd["model_param"])
    popt = torch.optim.Adam(pref.parameters())
    popt.load_state_dict(psd["optim"])
    for q in pref.parameters():      # every moment in the reference's shape, in the module's parameter order
        assert popt.state[q]["exp_avg"].shape == q.shape and popt.state[q]["exp_avg_sq"].shape == q.shape
    popt.param_groups[0]["lr"] = 1e-3
    ptr2 = Trainer(searched.SearchedNet(4, 6, 3, 2, 3, True, gene), graph=False)
    ck.load_train_state_dicts(ptr2, {"epoch": 3, "history": {}, "model_param": pref.state_dict(), "optim": popt.state_dict(),
                                     "scheduler": ck.scheduler_state_dict(ptr.scheduler), "best_loss": 0.4})
    assert int(ptr2.fp.step) == 5
    assert torch.equal(ptr2.fp.exp_avg, ptr.fp.exp_avg) and torch.equal(ptr2.fp.exp_avg_sq, ptr.fp.exp_avg_sq)
    assert torch.equal(ptr2.fp.flat, ptr.fp.flat)
    # genotype pickle (search.py:189-194, train.py:36-38)
    gp = tmp_path / "best_genotype.pkl"
    ck.save_genotype(gp, gene, 3)
    assert ck.load_genotype(gp) == gene


def test_channel_plan_matches_oracle_and_rejects_unsupported_widths():
    """unet.cell_specs: the (c0, c1, c_node, downward) plan equals the oracle's restatement of nas.py:33-49 / searched.py:74-90,
    and widths the 4-channel-vector kernels cannot run fail at construction with an explanation"""
    import pytest
    from nas_3d_unet_amd import unet
    from oracle import ref_path as orc
    for cfg in (orc.DEFAULT_CFG, orc.NetCfg(4, 8, 3, 3, 2, True), orc.NetCfg(1, 4, 1, 2, 4, False)):
        specs, head_in = unet.cell_specs(cfg.init_n_kernels, cfg.depth, cfg.n_nodes, cfg.channel_change)
        P = orc.supernet_param_specs(cfg)
        names = dict(P) if not isinstance(P, dict) else P
        for k, (c0, c1, cn, down) in enumerate(specs):
            pre = ("kernel.down_cells.%d." % k) if down else ("kernel.up_cells.%d." % (k - cfg.depth))
            assert tuple(names[pre + "preprocess0.conv.weight"])[:2] == (cn, c0), (k, specs[k])
            assert tuple(names[pre + "preprocess1.conv.weight"])[:2] == (cn, c1), (k, specs[k])
        assert tuple(names["kernel.last_conv.0.conv.weight"])[:2] == (cfg.out_channels, head_in)


def test_channel_counts_that_are_not_multiples_of_4_build_a_zero_padded_twin():
    """round 5: the reference takes any init_n_kernels (nas.py:13-26, searched.py:55-66).  Such a net keeps the reference's parameter
    names and shapes (checked against the oracle's inventory), and runs as a twin whose feature maps are zero-padded to multiples of 4:
    embedding the parameters and cutting them back is the identity, everything outside the real positions is zero, cell-output inputs
    are padded node by node and stem inputs at the end (host logic only: no kernel runs on the CPU)"""
    import torch
    from nas_3d_unet_amd import nas, searched, unet
    from oracle import ref_path as orc
    for init, nodes, depth, cc in ((2, 3, 3, True), (6, 3, 2, True), (5, 2, 2, False), (3, 3, 2, True)):
        assert unet.needs_padding(init, depth, nodes, cc)
        cfg = orc.NetCfg(4, init, 3, depth, nodes, cc)
        gene = orc.Genotype([("down_conv", 0), ("down_dep_conv", 1), ("down_se_conv", 0), ("dep_conv", 2), ("max_pool", 1), ("identity", 2)][:2 * nodes],
                            [("conv", 0), ("up_conv", 1), ("up_dep_conv", 1), ("se_conv", 2), ("identity", 0), ("up_se_conv", 1)][:2 * nodes])
        nets = [(searched.SearchedNet(4, init, 3, depth, nodes, cc, searched.Genotype(list(gene.down), list(gene.up))), orc.searched_param_specs(cfg, gene), ""),
                (nas.ShellNet(4, init, 3, depth, nodes, False, cc).kernel, orc.supernet_param_specs(cfg), "kernel.")]
        for net, specs, prefix in nets:
            want = {n[len(prefix):]: tuple(shape) for n, shape in specs if n.startswith(prefix) and "alpha" not in n}
            assert {n: tuple(p.shape) for n, p in net.named_parameters()} == want
            with torch.no_grad():
                for q in net.parameters():
                    q.copy_(torch.randn(q.shape))
            tw = net._n3d_make_twin()
            tw.embed(net)
            tp = dict(tw.twin.named_parameters())
            for n, r in net.named_parameters():
                t = tp[n].detach()
                assert torch.equal(tw.extract(n, t), r.detach()), n
                assert abs(float(t.abs().sum()) - float(r.detach().abs().sum())) <= 1e-4 * (1.0 + float(r.detach().abs().sum())), n
            # every feature-map width of the twin is a multiple of 4; the head still emits the real classes
            for name, m in tw.twin.named_modules():
                if isinstance(m, torch.nn.GroupNorm):
                    assert m.num_channels % 4 == 0 and 0 < m._n3d_real_c <= m.num_channels
            assert tp["last_conv.0.conv.weight"].shape[0] == 3
            # a cell-output input is padded node by node: real channel j of node k sits at k * pad4(c) + j
            w = tp["last_conv.0.conv.weight"].detach()
            c = net.last_conv[0].conv.weight.shape[1] // nodes
            cp = (c + 3) // 4 * 4
            assert w.shape[1] == nodes * cp
            assert torch.equal(w[:, [k * cp + j for k in range(nodes) for j in range(c)]], net.last_conv[0].conv.weight.detach())
            assert float(w[:, [k * cp + j for k in range(nodes) for j in range(c, cp)]].abs().sum()) == 0.0


def test_search_checkpoint_interop_with_torch_adam(tmp_path):
    """search.py:166-176 / 108-127: optim_shell / optim_kernel / both schedulers / geno_count round-trip through torch objects"""
    import torch
    from collections import Counter
    from nas_3d_unet_amd import checkpoint as ck, nas
    from nas_3d_unet_amd.train import SearchTrainer
    tr = SearchTrainer(nas.ShellNet(4, 4, 3, 2, 3, False, True), graph=False)    # construction only: no kernel runs on the CPU
    g = torch.Generator().manual_seed(0)
    tr.fp.exp_avg.copy_(torch.randn(tr.fp.numel, generator=g)); tr.fp.exp_avg_sq.copy_(torch.rand(tr.fp.numel, generator=g))
    tr.a_m.copy_(torch.randn(tr.a_m.numel(), generator=g)); tr.a_v.copy_(torch.rand(tr.a_v.numel(), generator=g))
    tr.fp.step.fill_(9); tr.a_step.fill_(9)
    tr.set_shell_lr(5e-4); tr.set_kernel_lr(2.5e-4)
    tr.shell_scheduler.num_bad_epochs, tr.kernel_scheduler.num_bad_epochs = 3, 5
    sd = ck.search_state_dicts(tr, epoch=4, geno_count=Counter({"g": 2}), history={"loss": [0.6]}, best_loss=0.4)
    assert set(sd) == {"epoch", "geno_count", "history", "model_param", "optim_shell", "optim_kernel", "kernel_scheduler", "shell_scheduler", "best_loss"}
    path = tmp_path / "last.pth"
    torch.save(sd, path)
    sd2 = torch.load(path, weights_only=False)
    ref = nas.ShellNet(4, 4, 3, 2, 3, False, True)
    ref.load_state_dict(sd2["model_param"])
    osh, okn = torch.optim.Adam(ref.alphas()), torch.optim.Adam(ref.kernel.parameters())
    osh.load_state_dict(sd2["optim_shell"]); okn.load_state_dict(sd2["optim_kernel"])
    assert osh.param_groups[0]["lr"] == 5e-4 and okn.param_groups[0]["lr"] == 2.5e-4
    a0 = list(ref.alphas())[1]
    o = tr.afp.offsets[1]
    assert torch.equal(osh.state[a0]["exp_avg"], tr.a_m[o:o + a0.numel()].view(a0.shape)) and float(osh.state[a0]["step"]) == 9
    tr2 = SearchTrainer(nas.ShellNet(4, 4, 3, 2, 3, False, True), graph=False)
    back = dict(sd2, optim_shell=osh.state_dict(), optim_kernel=okn.state_dict())
    epoch, gc_, _, best = ck.load_search_state_dicts(tr2, back)
    assert epoch == 5 and gc_["g"] == 2 and best == 0.4 and tr2.lr_shell == 5e-4 and tr2.lr_kernel == 2.5e-4
    assert int(tr2.a_step) == 9 and int(tr2.fp.step) == 9 and tr2.shell_scheduler.num_bad_epochs == 3 and tr2.kernel_scheduler.num_bad_epochs == 5
    for p, o in zip(tr.aparams, tr.afp.offsets):
        n = p.numel()
        assert torch.equal(tr2.a_m[o:o + n], tr.a_m[o:o + n]) and torch.equal(tr2.a_v[o:o + n], tr.a_v[o:o + n])
    assert torch.equal(tr2.fp.exp_avg, tr.fp.exp_avg) or all(
        torch.equal(tr2.fp.exp_avg[o:o + p.numel()], tr.fp.exp_avg[o:o + p.numel()]) for p, o in zip(tr.fp.params, tr.fp.offsets))


def test_genotype_file_is_parsed_not_evaluated(tmp_path):
    """a genotype file is data -- no class lookup while unpickling, no eval of its text"""
    import pickle
    import pytest
    from nas_3d_unet_amd import checkpoint as ck
    bad = tmp_path / "bad.pkl"
    with open(bad, "wb") as f:
        pickle.dump(("__import__('os').system('true')", 1), f)
    with pytest.raises(ValueError):
        ck.load_genotype(bad)
    with open(bad, "wb") as f:
        pickle.dump((pytest.raises, 1), f)     # needs a class / function lookup
    with pytest.raises(pickle.UnpicklingError):
        ck.load_genotype(bad)


def test_dropout_gate_host_generator_is_uniform():
    """the counter-based generator behind n3d_dropout3d_gate (host-callable twin n3d_dropout3d_uniform): values in [0, 1),
    deterministic, mean 1/2, different counters give different streams"""
    import ctypes as C
    import numpy as np
    from nas_3d_unet_amd import _lib
    lib = _lib.load()
    u0 = np.array([lib.n3d_dropout3d_uniform(C.c_uint64(77), 0, i) for i in range(4000)])
    u1 = np.array([lib.n3d_dropout3d_uniform(C.c_uint64(77), 1, i) for i in range(4000)])
    assert u0.min() >= 0.0 and u0.max() < 1.0 and abs(u0.mean() - 0.5) < 0.02 and abs((u0 < 0.1).mean() - 0.1) < 0.02
    assert not np.array_equal(u0, u1)
    assert lib.n3d_dropout3d_uniform(C.c_uint64(77), 0, 5) == u0[5]


def test_batch_job_tables_encode_scattered_addresses():
    """n3d_pack_batch / n3d_wgrad_finalize_batch compact their job tables into the kernel arguments (segment + offset);
    the host-side grouping must make progress and decode exactly for addresses scattered over terabytes."""
    from nas_3d_unet_amd import _lib
    launches = _lib.load().n3d_selftest_job_tables()
    assert 4 <= launches <= 300, launches
