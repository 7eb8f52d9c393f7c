"""prob_mbrl.models (models/__init__.py of the reference: everything of core / modules / densities at package
level, the three files as sub-modules)."""
from prob_mbrl_amd.models import *  # noqa: F401,F403
from prob_mbrl_amd.models import (BDropout, BSequential, CDropout, DiagGaussianDensity, DynamicsModel,  # noqa: F401
                                  GaussianMixtureDensity, Policy, Regressor, StochasticModule, mlp)

from . import core, densities, modules  # noqa: E402,F401
