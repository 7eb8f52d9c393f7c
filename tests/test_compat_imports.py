"""CPU: the reference's import surface resolves through compat/prob_mbrl (no device needed to import or to build the
modules)."""
import os
import sys

import numpy as np
import torch

from tests import common

for _p in (common.ROOT, os.path.join(common.ROOT, 'compat')):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def test_reference_import_lines_resolve():
    from prob_mbrl import utils, models, algorithms, envs, losses      # examples/deep_pilco_mm.py:11
    import prob_mbrl.models.modules as modules
    import prob_mbrl.models.core as core
    import prob_mbrl.models.densities as densities
    import prob_mbrl.utils.angles as angles
    import prob_mbrl_amd as pm
    assert modules.CDropout is pm.models.CDropout and modules.BDropout is pm.models.BDropout
    assert core.mlp is pm.models.mlp and core.DynamicsModel is pm.models.DynamicsModel and core.Policy is pm.models.Policy
    assert densities.DiagGaussianDensity is pm.models.DiagGaussianDensity
    assert densities.GaussianMixtureDensity is pm.models.GaussianMixtureDensity
    assert models.modules is modules and callable(algorithms.mc_pilco) and callable(utils.rollout)
    assert utils.load_csv('200,200') == [200, 200] and utils.load_csv('x') is None
    assert 'Cartpole' in envs.__all__ and callable(envs.__dict__['Cartpole'])
    assert envs.cartpole.CartpoleReward is pm.rewards.CartpoleReward
    assert callable(losses.gaussian_log_likelihood) and callable(angles.to_complex)
    for name in ('ExperienceDataset', 'SumTree', 'apply_controller', 'load_checkpoint', 'train_regressor', 'tile'):
        assert hasattr(utils, name), name


def test_reference_shaped_reward_is_adapted_at_construction():
    from prob_mbrl import models
    import prob_mbrl_amd as pm

    class PendulumReward(torch.nn.Module):          # envs/pendulum/env.py:27-40's attribute contract
        def __init__(self):
            super().__init__()
            P = lambda t: torch.nn.Parameter(t, requires_grad=False)  # noqa: E731
            self.Q, self.R = P(4.0 * torch.eye(2)), P(1e-4 * torch.eye(1))
            self.target, self.pole_length = P(torch.tensor([[float(np.pi), 0.0]])), P(torch.tensor(0.8))

    dyn = models.DynamicsModel(models.mlp(3, 4, [16, 16], dropout_layers=[models.modules.CDropout(0.1 * np.ones(16))] * 2,
                                          nonlin=torch.nn.ReLU), reward_func=PendulumReward(),
                               output_density=models.DiagGaussianDensity(2))
    assert isinstance(dyn.reward_func, pm.rewards.PendulumReward)
    sp = dyn.reward_func.spec(2)
    assert abs(sp['norm'] - 1.6) < 1e-6 and np.allclose(sp['Q'], 4.0 * np.eye(2))
