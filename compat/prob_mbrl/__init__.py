"""`import prob_mbrl` for scripts written against the reference: put `<repo>/compat` (and `<repo>`) on PYTHONPATH
and the reference's import lines resolve to this build --

    from prob_mbrl import utils, models, algorithms, envs, losses
    models.modules.CDropout / models.modules.BDropout / models.core.mlp / models.densities.DiagGaussianDensity ...
    utils.load_csv, utils.ExperienceDataset, utils.apply_controller, utils.train_regressor, utils.rollout ...
    envs.__all__, envs.Cartpole, envs.cartpole.CartpoleReward ...

-- so the construction code of examples/deep_pilco_mm.py:117-151 and the calls of :245-264 run as written
(examples/deep_pilco_mm_compat.py is that flow; tests/test_gpu_compat.py runs it).  The sub-packages mirror the
reference's module layout (models/{core,modules,densities}.py, utils/{core,angles,...}.py, envs/<env>/) as
namespaces over prob_mbrl_amd's implementations; nothing is implemented here."""
import importlib
import sys

import prob_mbrl_amd as _pkg  # noqa: F401

for _name in ('algorithms', 'rewards', 'losses'):
    _mod = importlib.import_module('prob_mbrl_amd.' + _name)
    sys.modules[__name__ + '.' + _name] = _mod
    globals()[_name] = _mod
from . import envs, models, utils  # noqa: E402,F401

__all__ = ['utils', 'models', 'algorithms', 'rewards', 'losses', 'envs']
