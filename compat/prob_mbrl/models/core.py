"""prob_mbrl.models.core (models/core.py: mlp :25-101, Regressor :104-187, Policy :190-248, DynamicsModel :251-303)."""
from prob_mbrl_amd.models import DynamicsModel, Policy, Regressor, mlp  # noqa: F401
