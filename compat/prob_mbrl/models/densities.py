"""prob_mbrl.models.densities (models/densities.py: DiagGaussianDensity :87-121, GaussianMixtureDensity :151-259)."""
from prob_mbrl_amd.models import DiagGaussianDensity, GaussianMixtureDensity  # noqa: F401
