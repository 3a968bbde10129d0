"""GPU: supernet search step (architecture pass on the validation batch, weight pass on the training batch,
search.py:211-238) against the CPU oracle driven by torch.optim.Adam, depth-2 supernet on 16^3 patches."""
import numpy as np
import pytest
import torch

from _util import fill_module
from oracle import ref_path as orc

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("graph", [False, True])
def test_search_step_matches_oracle(graph):
    from nas_3d_unet_amd import nas
    from nas_3d_unet_amd.train import SearchTrainer
    cfg = orc.DEFAULT_CFG._replace(depth=2)
    rng = np.random.default_rng(11)
    mk = lambda: (rng.standard_normal((2, 4, 16, 16, 16)).astype(np.float32),
                  (rng.uniform(0, 1, (2, 3, 16, 16, 16)) < 0.3).astype(np.float32))
    (xn, tn), (vxn, vtn) = mk(), mk()
    # ---- oracle: two Adam optimisers, alternating passes
    P = orc.make_params(orc.supernet_param_specs(cfg), requires_grad=True)
    alphas = [P[n] for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up")]
    kern = [v for n, v in P.items() if n.startswith("kernel.")]
    oa, ok = torch.optim.Adam(alphas), torch.optim.Adam(kern)
    ref = []
    for _ in range(2):
        oa.zero_grad()
        la = orc.dice_loss(orc.supernet_forward(P, torch.from_numpy(vxn), cfg), torch.from_numpy(vtn))
        la.backward(); oa.step()
        ok.zero_grad()
        lw = orc.dice_loss(orc.supernet_forward(P, torch.from_numpy(xn), cfg), torch.from_numpy(tn))
        lw.backward(); ok.step()
        ref.append((float(la), float(lw)))
    # ---- HIP
    net = nas.ShellNet(cfg.in_channels, cfg.init_n_kernels, cfg.out_channels, cfg.depth, cfg.n_nodes, False, cfg.channel_change)
    fill_module(net)
    net.kernel.last_conv[0].dropout = None
    net = net.cuda()
    tr = SearchTrainer(net, graph=graph)
    x, t, vx, vt = (torch.from_numpy(a).cuda() for a in (xn, tn, vxn, vtn))
    got = []
    for _ in range(2):
        la, lw = tr.step(x, t, vx, vt)
        got.append((float(la), float(lw)))
    np.testing.assert_allclose(np.array(got), np.array(ref), rtol=0, atol=3e-4)
    for n in ("alpha2_down", "alpha2_up", "alpha1_down", "alpha1_up"):
        mine = getattr(net, n).detach().cpu().numpy()
        # Adam moves each alpha by ~lr per step; compare with slack for sign flips of noise-level gradients
        assert np.abs(mine - P[n].detach().numpy()).max() <= 2.5e-3, n
    gene = net.get_gene()
    assert len(gene.down) == 6 and len(gene.up) == 6
