"""prob_mbrl.utils (utils/__init__.py of the reference re-exports core, rollout, experience_dataset,
apply_controller, train_regressor and the angles module)."""
from prob_mbrl_amd.utils import get_z_rnd, load_csv, rollout, tile, to_complex, train_regressor  # noqa: F401
from prob_mbrl_amd.experience import ExperienceDataset, SumTree, apply_controller, load_checkpoint  # noqa: F401

from . import angles  # noqa: E402,F401


def plot_rollout(*args, **kwargs):
    """utils/core.py plot_rollout / plot_trajectories: matplotlib drawing of sampled trajectories -- outside the
    rollout path this build covers (out of scope); the examples only call it with --plot_level > 0."""
    raise NotImplementedError('plotting is not part of the MI355X build (run the examples with --plot_level 0)')
