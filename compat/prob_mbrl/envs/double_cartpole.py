"""prob_mbrl.envs.double_cartpole: the analytic reward of envs/double_cartpole/env.py (the simulator itself is out of scope)."""
from prob_mbrl_amd.rewards import DoubleCartpoleReward  # noqa: F401
