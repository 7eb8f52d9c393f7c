"""prob_mbrl.envs.cart_acrobot: the analytic reward of envs/cart_acrobot/env.py (the simulator itself is out of scope)."""
from prob_mbrl_amd.rewards import CartAcrobotReward  # noqa: F401
