"""`import prob_mbrl` for scripts written against the reference: put `<repo>/compat` (and `<repo>`) on PYTHONPATH and
`from prob_mbrl import utils, models, algorithms` resolves to prob_mbrl_amd's modules of the same names (the
analytic reward classes the reference keeps under `prob_mbrl.envs.<env>` are `prob_mbrl.rewards.*` here)."""
import importlib
import sys

import prob_mbrl_amd as _pkg

for _name in ('utils', 'models', 'algorithms', 'rewards', 'losses', 'envs'):
    try:
        _mod = importlib.import_module('prob_mbrl_amd.' + _name)
    except ImportError:
        continue
    sys.modules[__name__ + '.' + _name] = _mod
    globals()[_name] = _mod
__all__ = [n for n in ('utils', 'models', 'algorithms', 'rewards', 'losses', 'envs') if n in globals()]
