"""prob_mbrl.utils.angles (utils/angles.py:7-42): [others | sin(angles) | cos(angles)]."""
from prob_mbrl_amd.utils import to_complex  # noqa: F401
