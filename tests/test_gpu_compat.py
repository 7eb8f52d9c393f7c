"""GPU (-m gpu): the reference's import surface (`compat/prob_mbrl`).  The construction code of the reference's
examples (examples/deep_pilco_mm.py:111-151: models.mlp with models.modules.CDropout / BDropout layers,
models.DynamicsModel(..., reward_func=env.reward_func, ...), models.Policy) written against `prob_mbrl`, with a
reward module shaped like the reference's own (a torch module named CartpoleReward holding Q, R, target, pole_length),
must reproduce the real reference's algorithms.mc_pilco run (fixtures mcp_*: losses of every iteration, final
parameters)."""
import os
import sys
from functools import partial

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
for _p in (common.ROOT, os.path.join(common.ROOT, 'compat')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


class CartpoleReward(torch.nn.Module):
    """The attribute contract of envs/cartpole/env.py:27-40 (its forward is never called by the rollout path)."""

    def __init__(self, pole_length=0.5):
        super().__init__()
        P = lambda t: torch.nn.Parameter(t, requires_grad=False)  # noqa: E731
        self.Q, self.R = P(16.0 * torch.eye(2)), P(1e-4 * torch.eye(1))
        self.target = P(torch.tensor([[0.0, 0.0, float(np.pi), 0.0]]))
        self.pole_length = P(torch.tensor(pole_length))


@pytest.mark.parametrize('name', ['mcp_mm1', 'mcp_nomm'])
def test_reference_construction_code_through_compat_reproduces_mc_pilco(name):
    from prob_mbrl import algorithms, models, utils          # the reference's import line
    import prob_mbrl.models.modules as ref_modules            # ... and its sub-module paths
    assert ref_modules.CDropout is models.modules.CDropout and models.core.mlp is models.mlp
    assert utils.load_csv('200,200') == [200, 200]
    d = common.load(name)
    B, D = d['x0'].shape
    U = d['pol_z'].shape[1]
    dyn_shape = [d['dyn_W%d' % i].shape[0] for i in range(int(d['dyn_n_layers']) - 1)]
    pol_shape = [d['pol_W%d' % i].shape[0] for i in range(int(d['pol_n_layers']) - 1)]
    maxU = np.asarray(d['pol_scale'], dtype=np.float32)
    dyn_model = models.mlp(D + U, 2 * D, dyn_shape,
                           dropout_layers=[models.modules.CDropout(0.1 * np.ones(hid)) for hid in dyn_shape],
                           nonlin=torch.nn.ReLU)
    dyn = models.DynamicsModel(dyn_model, reward_func=CartpoleReward(),
                               output_density=models.DiagGaussianDensity(D)).float()
    pol_model = models.mlp(D, 2 * U, pol_shape, dropout_layers=[models.modules.BDropout(0.1) for hid in pol_shape],
                           nonlin=torch.nn.ReLU, output_nonlin=partial(models.DiagGaussianDensity, U))
    pol = models.Policy(pol_model, maxU, -maxU).float()
    import prob_mbrl_amd as pm
    assert isinstance(dyn.reward_func, pm.rewards.CartpoleReward)      # restated from the module's constants
    common.fill_modules(dyn, pol, d)
    dyn, pol = dyn.to(DEV), pol.to(DEV)
    dyn.eval()
    opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
    gam = np.asarray(d['gamma'])
    losses = []
    algorithms.mc_pilco(torch.tensor(d['x0'], device=DEV), dyn, pol, int(d['H']), opt, None, int(d['mcp_n_iters']),
                        mm_states=bool(d['mm_states']), mm_rewards=bool(d['mm_rewards']),
                        mm_groups=int(d['mm_groups']) or None, maximize=True, clip_grad=float(d['mcp_clip']),
                        discount=None if np.allclose(gam, gam[0]) else float(gam[1] / gam[0]),
                        on_iteration=lambda i, loss, *a: losses.append(float(loss)),
                        frozen_noise=dict(z_mm=torch.tensor(d['z_mm']), z_rr=torch.tensor(d['z_rr'])))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
    lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
    final = torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()
    assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)


def test_reward_adapter_reads_the_modules_constants():
    import prob_mbrl_amd as pm
    r = CartpoleReward(pole_length=0.7)
    r.Q.data = 9.0 * torch.eye(2)
    ours = pm.rewards.from_module(r)
    sp = ours.spec(4)
    assert np.allclose(sp['Q'], 9.0 * np.eye(2)) and abs(sp['norm'] - 1.4) < 1e-6 and sp['expand']
    assert pm.rewards.from_module(ours) is ours and pm.rewards.from_module(torch.nn.Linear(2, 2)) is None


def test_compat_example_script_runs():
    """examples/deep_pilco_mm_compat.py end to end (experience -> BNN fit -> moment-matched policy search), small."""
    sys.path.insert(0, os.path.join(common.ROOT, 'examples'))
    import deep_pilco_mm_compat as ex
    dyn, pol, log = ex.main(['--ps_iters', '1', '--dyn_opt_iters', '30', '--pol_opt_iters', '5', '--pred_H', '8',
                             '--pol_batch_size', '30', '--dyn_shape', '32,32', '--pol_shape', '32,32',
                             '--reference_shaped_reward'])
    assert len(log) == 1 and len(log[0]['losses']) == 5 and all(np.isfinite(log[0]['losses']))
    assert all(torch.isfinite(p).all() for p in pol.parameters())
