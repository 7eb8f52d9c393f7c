"""Problem dictionaries: the flat, numpy description of one MC-PILCO rollout
problem (weights, normalisation, frozen masks / noise, reward constants) shared
by the benchmark, the tests and the golden fixtures (tools/make_golden.py keys).

`synthetic_problem` generates the benchmark inputs of SURVEY.md section 8(d) with
numpy only, so the GPU box needs neither the reference nor a dataset.
"""
import math

import numpy as np
import torch

# name -> (D, U, pol_hidden, dyn_hidden, particles, samples, H, mm, reward)
CONFIGS = {
    # BASELINE.json configs[1]: Cartpole-shaped, no moment matching
    'cartpole_nomm': dict(D=4, U=1, pol_hid=[200, 200], dyn_hid=[200, 200], P=100, S=25, H=40,
                          mm=False, reward='cartpole', maxU=10.0),
    # configs[2]: + moment matching per particle group
    'cartpole_mm': dict(D=4, U=1, pol_hid=[200, 200], dyn_hid=[200, 200], P=100, S=25, H=40,
                        mm=True, reward='cartpole', maxU=10.0),
    # configs[3]: Pendubot/Acrobot-shaped, per-GPU share of 400 x 50 over 4 GPUs
    'dcartpole_mm': dict(D=6, U=1, pol_hid=[200, 200], dyn_hid=[200, 200], P=100, S=50, H=60,
                         mm=True, reward='double_cartpole', maxU=20.0),
    # configs[4]: MFMA stress, per-GPU share of 2048 x 64 over 8 GPUs
    'stress32': dict(D=32, U=8, pol_hid=[512, 512, 512], dyn_hid=[512, 512, 512], P=256, S=64,
                     H=100, mm=False, reward='generic', maxU=1.0),
    # configs[4] as BASELINE.md's table has it: moment matching with mm_groups = particles (64-row groups, a 32 x 32
    # covariance per group and step)
    # (z_orth: see synthetic_problem -- i.i.d. moment-matching noise makes this row's 100-step rollout ill-conditioned
    #  beyond what the reference's own fp32 Cholesky survives)
    'stress32_mm': dict(D=32, U=8, pol_hid=[512, 512, 512], dyn_hid=[512, 512, 512], P=256, S=64,
                        H=100, mm=True, reward='generic', maxU=1.0, z_orth=True),
    # a 16-wide state with narrow networks: moment matching / span-form tests beyond the cart-pole widths
    'mid16_mm': dict(D=16, U=2, pol_hid=[64, 64], dyn_hid=[64, 64], P=4, S=576, H=4,
                     mm=True, reward='generic', maxU=1.0),
    # a small wide-network case for the parity tests of the general kernel family (hidden layers wide
    # enough that the K-split partial tiles live in the output buffer's free columns)
    'wide_small': dict(D=12, U=3, pol_hid=[272, 256], dyn_hid=[256, 288, 256], P=10, S=4, H=6,
                       mm=False, reward='generic', maxU=1.0),
    # action vectors wider than 8 (the reward launch's 16-wide instance of the action cost) / between 5 and 8
    'wide_actions': dict(D=10, U=12, pol_hid=[64, 64], dyn_hid=[64, 64], P=6, S=8, H=6,
                         mm=False, reward='generic', maxU=1.0),
    'mid_actions': dict(D=10, U=6, pol_hid=[64, 64], dyn_hid=[64, 64], P=6, S=8, H=6,
                        mm=False, reward='generic', maxU=1.0),
}


def cartpole_reward_spec(D, l=0.5, Q=16.0, R=1e-4):
    """envs/cartpole/env.py:27-86 restated as the generic spec; raw 4-D state
    (angle at index 2 expanded inside the reward) or 5-D expanded state."""
    C = np.zeros((2, 5))
    C[0, 0], C[0, 3], C[1, 4] = 1.0, l, -l
    targeta = np.array([0.0, 0.0, 0.0, np.sin(np.float32(np.pi)), np.cos(np.float32(np.pi))])
    return dict(kind='exp', expand=(D == 4), angle_dims=[2], C=C, tip_target=C @ targeta,
                norm=2 * l, w=0.5, Q=Q * np.eye(2), R=R * np.eye(1))


def double_cartpole_reward_spec(D, l1=0.6, l2=0.6, Q=8.0, R=1e-3):
    """envs/double_cartpole/env.py:27-91; raw 6-D state, angles at 2 and 4."""
    C = np.zeros((2, 8))
    C[0, 0], C[0, 4], C[0, 5] = 1.0, -l1, -l2
    C[1, 6], C[1, 7] = l1, l2
    targeta = np.array([0, 0, 0, 0, 0, 0, 1.0, 1.0])
    return dict(kind='exp', expand=(D == 6), angle_dims=[2, 4], C=C, tip_target=C @ targeta,
                norm=2 * (l1 + l2), w=0.5, Q=Q * np.eye(2), R=R * np.eye(1))


def generic_reward_spec(D, U, rng, k=2):
    """exp(-1/2 ||C x - c||^2_Q) with a fixed random k x D map."""
    C = rng.standard_normal((k, D)) / math.sqrt(D)
    return dict(kind='exp', expand=False, angle_dims=[], C=C, tip_target=np.zeros(k), norm=1.0,
                w=0.5, Q=np.eye(k), R=1e-3 * np.eye(U))


def _xavier(rng, out_f, in_f):
    # models/core.py:20-22: xavier_normal_ with relu gain, biases U(-0.1, 0.1)
    std = math.sqrt(2.0) * math.sqrt(2.0 / (in_f + out_f))
    return (rng.standard_normal((out_f, in_f)) * std).astype(np.float32), \
        rng.uniform(-0.1, 0.1, out_f).astype(np.float32)


def synthetic_problem(name='cartpole_nomm', seed=0, P=None, S=None, H=None, data_seed=None):
    cfg = dict(CONFIGS[name])
    if P is not None:
        cfg['P'] = P
    if S is not None:
        cfg['S'] = S
    if H is not None:
        cfg['H'] = H
    D, U, Pn, Sn, Hn = cfg['D'], cfg['U'], cfg['P'], cfg['S'], cfg['H']
    B = Pn * Sn
    rng = np.random.default_rng(seed)
    d = {}
    pol_dims = [D] + cfg['pol_hid'] + [2 * U]
    dyn_dims = [D + U] + cfg['dyn_hid'] + [2 * D]
    for pre, dims in (('pol', pol_dims), ('dyn', dyn_dims)):
        d[pre + '_n_layers'] = len(dims) - 1
        for i in range(len(dims) - 1):
            W, b = _xavier(rng, dims[i + 1], dims[i])
            d['%s_W%d' % (pre, i)] = W
            d['%s_b%d' % (pre, i)] = b
    # normalisation from a synthetic dataset X ~ N(0,1), Y ~ 0.01 N(0,1) (models/core.py:134-149)
    X = cfg.get('x_scale', 1.0) * rng.standard_normal((300, D + U))
    Y = cfg.get('y_scale', 0.01) * rng.standard_normal((300, D))
    rew_rng = np.random.default_rng([seed, 12345])
    if data_seed is not None:   # per-rank particles / masks / noise; weights stay shared
        rng = np.random.default_rng([seed, data_seed])
    for pre, dims in (('pol', pol_dims), ('dyn', dyn_dims)):
        for i in range(len(dims) - 2):
            d['%s_mask%d' % (pre, i)] = (rng.random((B, dims[i + 1])) < 0.9).astype(np.float32)
    d['pol_keep'] = np.full(len(pol_dims) - 2, np.float32(0.9), dtype=np.float32)   # BDropout
    d['dyn_keep'] = np.ones(len(dyn_dims) - 2, dtype=np.float32)                    # CDropout
    d['pol_z'] = rng.standard_normal((B, U)).astype(np.float32)
    d['dyn_z'] = rng.standard_normal((B, D)).astype(np.float32)
    maxU = np.full(U, cfg['maxU'], dtype=np.float32)
    d['pol_scale'], d['pol_bias'] = maxU, np.zeros(U, dtype=np.float32)
    d['pol_angle_dims'] = np.zeros(0, dtype=np.int64)
    d['dyn_angle_dims'] = np.zeros(0, dtype=np.int64)
    d['dyn_mx'] = X.mean(0).astype(np.float32)
    Sx = (4.0 * X.std(0, ddof=1)).astype(np.float32)
    d['dyn_iSx'] = (1.0 / Sx).astype(np.float32)
    d['dyn_my'] = Y.mean(0).astype(np.float32)
    d['dyn_Sy'] = (4.0 * Y.std(0, ddof=1)).astype(np.float32)
    if cfg['reward'] == 'cartpole':
        spec = cartpole_reward_spec(D)
    elif cfg['reward'] == 'double_cartpole':
        spec = double_cartpole_reward_spec(D)
    else:
        spec = generic_reward_spec(D, U, rew_rng)
    d['rew_kind'], d['rew_expand'] = spec['kind'], spec['expand']
    d['rew_angle_dims'] = np.asarray(spec['angle_dims'], dtype=np.int64)
    d['rew_C'], d['rew_tip_target'] = spec['C'], spec['tip_target']
    d['rew_norm'], d['rew_w'], d['rew_Q'], d['rew_R'] = spec['norm'], spec['w'], spec['Q'], spec['R']
    x0p = (0.1 * rng.standard_normal((Pn, D))).astype(np.float32)
    d['x0'] = np.repeat(x0p, Sn, axis=0)        # utils.tile layout: row = p*S + s
    d['H'] = Hn
    d['gamma'] = np.full(Hn, 1.0 / Hn)
    d['mm_states'] = d['mm_rewards'] = bool(cfg['mm'])
    d['mm_groups'] = Pn if cfg['mm'] else 0
    d['maximize'] = True
    d['infer_ns'] = False
    d['z_mm'] = rng.standard_normal((Hn + B, D)).astype(np.float32)
    d['z_rr'] = rng.standard_normal((Hn + B, 1)).astype(np.float32)
    if cfg.get('z_orth'):
        # Orthogonalised moment-matching noise: rows periodic in the group size M, any M consecutive rows have zero
        # column means and the identity as sample covariance -- so mm_resample_'s m + z L^T reproduces the fitted
        # covariance exactly.  (With i.i.d. rows the cyclic index of utils/rollout.py:53-59 hands a group nearly the
        # SAME z block at consecutive steps; its sample correlation matrix C, eigenvalues (1 +- sqrt(D / M))^2, is
        # then applied again and again -- cov <- L C L^T -- and at D = 32, M = 64 the covariance's condition number
        # passes 1e14 within 50 steps: the reference's fp32 Cholesky raises after about 20, a fp64 one on fp32 states
        # after about 45.  A designed z keeps the H = 100 rollout of the C5 row well-conditioned; the arithmetic the
        # kernels do is the same.)
        M = Sn
        assert D + 1 < M
        A = np.concatenate([np.ones((M, 1)), rng.standard_normal((M, D + 1))], 1)
        Q = np.linalg.qr(A)[0][:, 1:] * math.sqrt(M - 1.0)
        reps = (Hn + B + M - 1) // M
        d['z_mm'] = np.tile(Q[:, :D], (reps, 1))[:Hn + B].astype(np.float32)
        d['z_rr'] = np.tile(Q[:, D:D + 1], (reps, 1))[:Hn + B].astype(np.float32)
    return d


def shard_problem(d, rank, world):
    """Rows [rank*B/world, (rank+1)*B/world) of problem d as one rank's problem (whole particle /
    moment-matching groups per rank); shared inputs (weights, normalisation, the cyclic
    moment-matching noise, which the kernels index with the global row) stay whole."""
    B = d['x0'].shape[0]
    assert B % world == 0
    per = B // world
    lo, hi = rank * per, (rank + 1) * per
    G = int(d['mm_groups'])
    assert G % world == 0 or G == 0, 'whole moment-matching groups per rank'
    out = dict(d)
    for k, v in d.items():
        v = np.asarray(v) if not isinstance(v, (str, bool, int, float)) else v
        if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and getattr(v, 'ndim', 0) == 2 and v.shape[0] == B):
            out[k] = v[lo:hi]
    out['mm_groups'] = G // world
    return out


# ---------------------------------------------------------------------------
def reward_spec_from_problem(d):
    return dict(kind=str(d['rew_kind']), expand=bool(d['rew_expand']),
                angle_dims=[int(a) for a in np.asarray(d['rew_angle_dims'])],
                C=np.asarray(d['rew_C']), tip_target=np.asarray(d['rew_tip_target']),
                norm=float(d['rew_norm']), w=float(d['rew_w']), Q=np.asarray(d['rew_Q']),
                R=np.asarray(d['rew_R']))


def flat_params(d, prefix):
    parts = []
    for i in range(int(d[prefix + '_n_layers'])):
        parts.append(np.asarray(d['%s_W%d' % (prefix, i)], dtype=np.float32).reshape(-1))
        parts.append(np.asarray(d['%s_b%d' % (prefix, i)], dtype=np.float32).reshape(-1))
    return np.concatenate(parts)


def layer_dims(d, prefix):
    n = int(d[prefix + '_n_layers'])
    return [d[prefix + '_W0'].shape[1]] + [d['%s_W%d' % (prefix, i)].shape[0] for i in range(n)]


def mlp_macs(dims):
    return sum(a * b for a, b in zip(dims[:-1], dims[1:]))


def algorithmic_flops_per_rollout(d):
    """2 H (3 P_macs + 2 F_macs) -- policy fwd + dX + dW, dynamics fwd + dX."""
    P = mlp_macs(layer_dims(d, 'pol'))
    F = mlp_macs(layer_dims(d, 'dyn'))
    return 2.0 * int(d['H']) * (3 * P + 2 * F), P, F


def loss_weights(d, B_global):
    """dL/dr[t,b] of algorithms/mc_pilco.py:134-144,190 (mean over the GLOBAL batch)."""
    sign = -1.0 if bool(d['maximize']) else 1.0
    g = np.asarray(d['gamma'], dtype=np.float64)
    return (sign * g[:, None] * np.ones((1, B_global)) / B_global).astype(np.float32)


def engine_from_problem(d, device='cuda:0', rows_per_wg_hint=0, shard=None, B_global=None,
                        row_offset=None, force_generic=False, no_shaped=False, precision=None, mm_span=None):
    """Build an Engine + its device input tensors from a problem dict.
    shard=(rank, world): this rank's contiguous block of rows (whole mm groups) of ONE
    global batch described by d.  B_global/row_offset instead place the whole of d as a
    shard of a larger global batch (weak scaling)."""
    from . import engine as E
    dev = torch.device(device)
    B, D = d['x0'].shape
    U = d['pol_z'].shape[1]
    H = int(d['H'])
    npl, ndl = int(d['pol_n_layers']), int(d['dyn_n_layers'])
    G = int(d['mm_groups'])
    lo, hi = 0, B
    Bg, roff = B, 0
    if shard is not None:
        rank, world = shard
        per = B // world
        lo, hi = rank * per, (rank + 1) * per
        roff = lo
    if B_global is not None:
        Bg, roff = B_global, row_offset
    Bl = hi - lo
    Gl = (G * Bl // B) if G > 0 else None
    # [H, B, h] masks: a fresh dropout mask at every step (resample_policy / resample_model=True)
    pol_ps = np.asarray(d['pol_mask0']).ndim == 3
    dyn_ps = np.asarray(d['dyn_mask0']).ndim == 3
    eng = E.Engine(Bl, D, U, H, layer_dims(d, 'pol'), list(np.asarray(d['pol_keep'])),
                   layer_dims(d, 'dyn'), list(np.asarray(d['dyn_keep'])),
                   reward_spec_from_problem(d), mm_states=bool(d['mm_states']),
                   mm_rewards=bool(d['mm_rewards']), mm_groups=Gl, device=dev, B_global=Bg,
                   row_offset=roff, rows_per_wg_hint=rows_per_wg_hint, force_generic=force_generic,
                   no_shaped=no_shaped, infer_ns=bool(d['infer_ns']) if 'infer_ns' in d else False,
                   precision=precision, pol_masks_per_step=pol_ps, dyn_masks_per_step=dyn_ps,
                   pol_angle_dims=[int(a) for a in np.asarray(d.get('pol_angle_dims', []))],
                   dyn_angle_dims=[int(a) for a in np.asarray(d.get('dyn_angle_dims', []))],
                   dyn_components=int(d['dyn_gmm_n']) if 'dyn_gmm_n' in d else 0, mm_span=mm_span)
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)  # noqa: E731

    def rows(m):
        m = np.asarray(m)
        if m.ndim == 3:      # per-step masks: [H, B, h] -> bit rows [H * B, nt]
            return T(m[:, lo:hi]).reshape(-1, m.shape[-1]).contiguous()
        return T(m[lo:hi])

    args = dict(
        x0=T(d['x0'][lo:hi]), pol_flat=T(flat_params(d, 'pol')), dyn_flat=T(flat_params(d, 'dyn')),
        mx=T(d['dyn_mx']), iSx=T(d['dyn_iSx']), my=T(d['dyn_my']), Sy=T(d['dyn_Sy']),
        pol_scale=T(d['pol_scale']), pol_bias=T(d['pol_bias']),
        pol_mask_bits=[E.pack_mask(rows(d['pol_mask%d' % i])) for i in range(npl - 1)],
        dyn_mask_bits=[E.pack_mask(rows(d['dyn_mask%d' % i])) for i in range(ndl - 1)],
        z_pol=T(d['pol_z'][lo:hi]),
        z_dyn=T(np.asarray(d['dyn_z'])[:, lo:hi] if np.asarray(d['dyn_z']).ndim == 3 else d['dyn_z'][lo:hi]),
        z_mm=T(d['z_mm']) if 'z_mm' in d and bool(d['mm_states']) else None,
        z_rr=T(d['z_rr']) if 'z_rr' in d and bool(d['mm_rewards']) else None)
    if 'dyn_gmm_n' in d and int(d['dyn_gmm_n']) > 1:
        # mixture head: frozen Gumbel noise and the uniforms of the per-step component draws ('dyn_ucat' [H, B])
        args.update(z_pi=T(np.asarray(d['dyn_zpi'])[lo:hi]), u_cat=T(np.asarray(d['dyn_ucat'])[:, lo:hi]))
    return eng, args, (lo, hi)
