"""prob_mbrl.models.modules (models/modules.py: StochasticModule, BDropout :19-61, CDropout :64-160, BSequential)."""
from prob_mbrl_amd.models import BDropout, BSequential, CDropout, StochasticModule  # noqa: F401
