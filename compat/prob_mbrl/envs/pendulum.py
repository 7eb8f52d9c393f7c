"""prob_mbrl.envs.pendulum: the analytic reward of envs/pendulum/env.py (the simulator itself is out of scope)."""
from prob_mbrl_amd.rewards import PendulumReward  # noqa: F401
