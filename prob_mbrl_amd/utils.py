"""`prob_mbrl.utils` names used on the MC-PILCO path."""
import torch

from .rollout import get_z_rnd, rollout  # noqa: F401


def tile(tensor, n):
    """[G, D] -> [G*n, D] with row g*n+k = row g (utils/core.py:188-190): the
    particle x sample layout (row = particle*S + sample) every kernel assumes."""
    return tensor.repeat_interleave(int(n), dim=0)


def to_complex(x, dims):
    """utils/angles.py:7-42: [others, sin(angles), cos(angles)]."""
    dims = [int(d) for d in dims]
    if len(dims) == 0:
        return x
    others = [i for i in range(x.shape[-1]) if i not in dims]
    return torch.cat([x[..., others], x[..., dims].sin(), x[..., dims].cos()], -1)


def load_csv(s):
    """utils/core.py:193-197: "200,200" -> [200, 200] (None if it does not parse) -- the argparse type of the
    examples' --dyn_shape / --pol_shape / --timesteps_to_sample."""
    try:
        return [int(d) for d in s.split(',')]
    except Exception:
        return None


def train_regressor(*args, **kwargs):
    """utils/train_regressor.py:58-165 (see prob_mbrl_amd/train_regressor.py)."""
    from .train_regressor import train_regressor as _tr
    return _tr(*args, **kwargs)


def __getattr__(name):
    # host data path (prob_mbrl_amd/experience.py), resolved lazily: it imports this module
    if name in ('ExperienceDataset', 'SumTree', 'apply_controller', 'load_checkpoint'):
        from . import experience
        return getattr(experience, name)
    raise AttributeError(name)
