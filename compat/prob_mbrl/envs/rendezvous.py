"""prob_mbrl.envs.rendezvous: the analytic reward of envs/rendezvous/env.py (the simulator itself is out of scope)."""
from prob_mbrl_amd.rewards import RendezvousReward  # noqa: F401
