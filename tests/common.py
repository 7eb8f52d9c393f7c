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
        return [n for n in names if not n.startswith(('mcp_', 'ext_', 'standalone_', 'bnn_', 'experience_', 'critic_', 'bnnopt_',
                                                      'trunc_', 'draw_', 'unit_'))]
    if kind == 'bnn':
        return [n for n in names if n.startswith('bnn_')]
    if kind == 'standalone':
        return [n for n in names if n.startswith('standalone_')]
    if kind == 'mcp':
        return [n for n in names if n.startswith('mcp_')]
    return names


def load(name):
    from oracle import ref_torch
    d = ref_torch.load_fixture(os.path.join(GOLDEN, name + '.npz'))
    if 'dyn_gmm_n' in d and int(d['dyn_gmm_n']) > 1 and 'dyn_ucat' not in d:
        d = dict(d)
        d['dyn_ucat'] = gmm_uniforms(d)
    return d


def gmm_uniforms(d):
    """Mixture head: the reference draws the component of every (step, row) from torch's generator and the fixture
    records the INDEX; the device draws by inverse CDF from a uniform.  The uniforms that reproduce the recorded
    indices: midpoints of the drawn component's CDF interval, the softmax taken from the fp64 oracle run."""
    import torch
    from oracle import ref_torch as R
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    dyn['k_soft_out'] = []
    R.iteration(x0, pol, dyn, spec, meta['H'], gamma, meta['maximize'], meta['mm_states'], meta['mm_rewards'],
                meta['mm_groups'], z_mm, z_rr, meta['infer_ns'])
    ks = torch.stack(dyn['k_soft_out'][:meta['H']]).numpy()          # [H, B, n]
    kidx = np.asarray(d['dyn_kidx'])
    cdf = np.cumsum(ks, -1)
    hi = np.take_along_axis(cdf, kidx[..., None], -1)[..., 0]
    lo = hi - np.take_along_axis(ks, kidx[..., None], -1)[..., 0]
    assert np.all(hi - lo > 1e-4), 'a drawn component has (almost) no probability: midpoint not robust'
    return np.clip(0.5 * (lo + hi), 0.0, 1.0 - 1e-7).astype(np.float32)


from prob_mbrl_amd.problem import engine_from_problem as engine_from_fixture  # noqa: E402,F401
from prob_mbrl_amd.problem import loss_weights  # noqa: E402,F401


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def modules_from_fixture(d, name, device='cuda:0'):
    """Build prob_mbrl_amd Policy / DynamicsModel (reference-shaped modules) holding the
    fixture's weights, normalisation, frozen masks and noise."""
    from functools import partial

    import prob_mbrl_amd as pm
    dev = torch.device(device)
    B, D = d['x0'].shape
    U = d['pol_z'].shape[1]
    npl, ndl = int(d['pol_n_layers']), int(d['dyn_n_layers'])
    pol_hid = [d['pol_W%d' % i].shape[0] for i in range(npl - 1)]
    dyn_hid = [d['dyn_W%d' % i].shape[0] for i in range(ndl - 1)]
    pad = [int(a) for a in np.asarray(d['pol_angle_dims'])]
    dad = [int(a) for a in np.asarray(d['dyn_angle_dims'])]
    if name.startswith(('dcp', 'angles_dcp', 'gmm_d6')):
        rew = pm.rewards.DoubleCartpoleReward(pole1_length=torch.tensor(0.6),
                                              pole2_length=torch.tensor(0.6))
    elif name.startswith('pend'):
        rew = pm.rewards.PendulumReward(pole_length=torch.tensor(1.0))
    elif name.startswith('rdv') or name.endswith('_u4'):
        rew = pm.rewards.RendezvousReward()
    elif name.startswith('c5'):
        rew = pm.rewards.LinearFeatureReward(torch.tensor(np.asarray(d['rew_C'], dtype=np.float32)),
                                             torch.tensor(np.asarray(d['rew_tip_target'], dtype=np.float32)),
                                             torch.tensor(np.asarray(d['rew_Q'], dtype=np.float32)),
                                             torch.tensor(np.asarray(d['rew_R'], dtype=np.float32)))
    else:
        rew = pm.rewards.CartpoleReward(pole_length=torch.tensor(0.5))
    n_comp = int(d['dyn_gmm_n']) if 'dyn_gmm_n' in d else 0
    dyn = pm.models.DynamicsModel(
        pm.models.mlp(D + U + len(dad), (2 * D + 1) * n_comp + 1 if n_comp > 1 else 2 * D, dyn_hid,
                      dropout_layers=[pm.models.CDropout(0.1 * np.ones(h)) for h in dyn_hid],
                      nonlin=torch.nn.ReLU),
        reward_func=rew, angle_dims=dad,
        output_density=(pm.models.GaussianMixtureDensity(D, n_comp) if n_comp > 1
                        else pm.models.DiagGaussianDensity(D))).float()
    maxU = np.asarray(d['pol_scale'], dtype=np.float32)
    pol = pm.models.Policy(
        pm.models.mlp(D + len(pad), 2 * U, pol_hid,
                      # (per-unit rates where the fixture recorded them: models/modules.py:19-27 takes a tensor)
                      dropout_layers=[pm.models.BDropout(torch.tensor(np.asarray(d['pol_rate%d' % i], dtype=np.float32))
                                                         if 'pol_rate%d' % i in d else 0.1)
                                      for i in range(len(pol_hid))],
                      nonlin=torch.nn.ReLU,
                      output_nonlin=partial(pm.models.DiagGaussianDensity, U)), maxU, -maxU, angle_dims=pad).float()
    fill_modules(dyn, pol, d)
    dyn = dyn.to(dev)
    pol = pol.to(dev)
    dyn.eval()
    return dyn, pol


def fill_modules(dyn, pol, d):
    """Copy the fixture's weights, normalisation, frozen masks and noise into reference-shaped modules (on the CPU,
    before they are moved to the device)."""
    import prob_mbrl_amd as pm
    B, D = d['x0'].shape
    npl, ndl = int(d['pol_n_layers']), int(d['dyn_n_layers'])
    dyn_hid = [d['dyn_W%d' % i].shape[0] for i in range(ndl - 1)]
    n_comp = int(d['dyn_gmm_n']) if 'dyn_gmm_n' in d else 0
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    with torch.no_grad():
        for pre, mod, n in (('pol', pol.model, npl), ('dyn', dyn.model, ndl)):
            lins = [m for m in mod._modules.values() if isinstance(m, torch.nn.Linear)]
            for i, lin in enumerate(lins):
                lin.weight.copy_(T(d['%s_W%d' % (pre, i)]))
                lin.bias.copy_(T(d['%s_b%d' % (pre, i)]))
        for i in range(npl - 1):
            getattr(pol.model, 'drop%d' % i).noise.data = T(d['pol_mask%d' % i])
        for i in range(ndl - 1):
            dr = getattr(dyn.model, 'drop%d' % i)
            dr.noise.data = torch.rand(B, dyn_hid[i])
            dr.concrete_noise = T(d['dyn_mask%d' % i])
        pol.model.fc_nonlin.z.data = T(d['pol_z'])
        if n_comp > 1:    # mixture head: frozen Gumbel noise; the per-step draws are replayed (rollout.Bundle)
            dyn.output_density.z_pi.data = T(d['dyn_zpi'])
            dyn.output_density._forced_draws = (T(d['dyn_z']), T(d['dyn_ucat']))
        else:
            dyn.output_density.z.data = T(d['dyn_z'])
        for k in ('mx', 'iSx', 'my', 'Sy'):
            getattr(dyn, k).data = T(d['dyn_' + k]).reshape(1, -1)
        dyn.Sx.data = dyn.iSx.reciprocal()
