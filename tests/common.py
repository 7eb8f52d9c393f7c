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


def reward_spec_from_fixture(d):
    return dict(kind=str(d['rew_kind']), expand=bool(d['rew_expand']),
                angle_dims=[int(a) for a in np.asarray(d['rew_angle_dims'])],
                C=np.asarray(d['rew_C']), tip_target=np.asarray(d['rew_tip_target']),
                norm=float(d['rew_norm']), w=float(d['rew_w']), Q=np.asarray(d['rew_Q']),
                R=np.asarray(d['rew_R']))


def flat_params(d, prefix, n_layers):
    parts = []
    for i in range(n_layers):
        parts.append(np.asarray(d['%s_W%d' % (prefix, i)], dtype=np.float32).reshape(-1))
        parts.append(np.asarray(d['%s_b%d' % (prefix, i)], dtype=np.float32).reshape(-1))
    return np.concatenate(parts)


def engine_from_fixture(d, device='cuda:0', rows_per_wg_hint=0, shard=None):
    """Build a prob_mbrl_amd Engine + its input tensors from a fixture dict.
    shard=(rank, world): rows split contiguously (whole mm groups)."""
    from prob_mbrl_amd import engine as E
    dev = torch.device(device)
    B, D = d['x0'].shape
    U = d['pol_z'].shape[1]
    H = int(d['H'])
    npl, ndl = int(d['pol_n_layers']), int(d['dyn_n_layers'])
    pol_dims = [d['pol_W0'].shape[1]] + [d['pol_W%d' % i].shape[0] for i in range(npl)]
    dyn_dims = [d['dyn_W0'].shape[1]] + [d['dyn_W%d' % i].shape[0] for i in range(ndl)]
    G = int(d['mm_groups'])
    lo, hi = 0, B
    if shard is not None:
        rank, world = shard
        per = B // world
        lo, hi = rank * per, (rank + 1) * per
    Bl = hi - lo
    Gl = (G * Bl // B) if G > 0 else None
    eng = E.Engine(Bl, D, U, H, pol_dims, list(np.asarray(d['pol_keep'])), dyn_dims,
                   list(np.asarray(d['dyn_keep'])), reward_spec_from_fixture(d),
                   mm_states=bool(d['mm_states']), mm_rewards=bool(d['mm_rewards']),
                   mm_groups=Gl, device=dev, B_global=B, row_offset=lo,
                   rows_per_wg_hint=rows_per_wg_hint)
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)  # noqa: E731
    args = dict(
        x0=T(d['x0'][lo:hi]), pol_flat=T(flat_params(d, 'pol', npl)),
        dyn_flat=T(flat_params(d, 'dyn', ndl)), mx=T(d['dyn_mx']), iSx=T(d['dyn_iSx']),
        my=T(d['dyn_my']), Sy=T(d['dyn_Sy']), pol_scale=T(d['pol_scale']),
        pol_bias=T(d['pol_bias']),
        pol_mask_bits=[E.pack_mask(T(d['pol_mask%d' % i][lo:hi])) for i in range(npl - 1)],
        dyn_mask_bits=[E.pack_mask(T(d['dyn_mask%d' % i][lo:hi])) for i in range(ndl - 1)],
        z_pol=T(d['pol_z'][lo:hi]), z_dyn=T(d['dyn_z'][lo:hi]),
        z_mm=T(d['z_mm']) if 'z_mm' in d else None,
        z_rr=T(d['z_rr']) if 'z_rr' in d else None)
    return eng, args, (lo, hi)


def loss_weights(d, B_global):
    """dL/dr[t,b] of algorithms/mc_pilco.py:134-144,190."""
    sign = -1.0 if bool(d['maximize']) else 1.0
    g = np.asarray(d['gamma'], dtype=np.float64)
    return (sign * g[:, None] * np.ones((1, B_global)) / B_global).astype(np.float32)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
