"""Shared helpers for the parity tests: fixture -> Engine inputs."""
import glob
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def fixture_names(kind=None):
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))
    if kind == 'iter':
        return [n for n in names if not n.startswith('mcp_')]
    if kind == 'mcp':
        return [n for n in names if n.startswith('mcp_')]
    return names


def load(name):
    from oracle import ref_torch
    return ref_torch.load_fixture(os.path.join(GOLDEN, name + '.npz'))


from prob_mbrl_amd.problem import engine_from_problem as engine_from_fixture  # noqa: E402,F401
from prob_mbrl_amd.problem import loss_weights  # noqa: E402,F401


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
