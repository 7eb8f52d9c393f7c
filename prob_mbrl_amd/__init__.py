"""prob_mbrl_amd -- MI355X-native MC-PILCO rollout + policy-gradient hot path with
the call surface of mcgillmrl/prob_mbrl:

    from prob_mbrl_amd import models, utils, algorithms, rewards
    dyn = models.DynamicsModel(models.mlp(D + U, 2 * D, [200, 200], dropout_layers=[...]),
                               reward_func=rewards.CartpoleReward(...),
                               output_density=models.DiagGaussianDensity(D)).float().cuda()
    pol = models.Policy(models.mlp(D, 2 * U, [200, 200], ...,
                                   output_nonlin=partial(models.DiagGaussianDensity, U)),
                        maxU, minU).float().cuda()
    algorithms.mc_pilco(x0, dyn, pol, H, opt, exp, ...)        # reference signature
    states, actions, rewards = utils.rollout(x0, dyn, pol, H)  # reference signature

The arithmetic runs in hand-written HIP kernels (prob_mbrl_amd/csrc) behind the C ABI
of include/pmbrl.h; there is no CPU or eager-torch fallback.
"""
from . import models, rewards, utils, algorithms, envs, losses  # noqa: F401
from .models import (BDropout, BSequential, CDropout, DiagGaussianDensity, DynamicsModel,  # noqa: F401
                     GaussianMixtureDensity, Policy, Regressor, mlp)

__all__ = ['models', 'rewards', 'utils', 'algorithms', 'envs', 'losses']
