"""prob_mbrl.envs.  The reference's environments are gym / Box2D wrappers around ODE simulators (envs/base.py,
envs/<env>/model.py) and are out of scope; what the rollout path takes from them is the analytic reward
module of each (`env.reward_func`, envs/<env>/env.py).  This namespace offers those reward classes under the
reference's names, plus the self-contained numpy cart-pole of prob_mbrl_amd.envs so that
`envs.__dict__[args.env]()` (examples/deep_pilco_mm.py:69-70) works for the default `--env Cartpole`."""
from prob_mbrl_amd.envs import Cartpole  # noqa: F401

from . import cart_acrobot, cartpole, double_cartpole, pendulum, rendezvous  # noqa: E402,F401

__all__ = ['Cartpole', 'cartpole', 'pendulum', 'double_cartpole', 'cart_acrobot', 'rendezvous']
