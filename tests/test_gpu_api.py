"""GPU (-m gpu): the reference-shaped Python surface (models / utils.rollout /
algorithms.mc_pilco) on top of the C ABI.  These read like the reference's own
usage: build modules, call rollout(), loss.backward(), mc_pilco()."""
import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _flat_grad(pol):
    lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
    return torch.cat([t.grad.reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()


@pytest.mark.parametrize('name', [n for n in common.fixture_names('iter') if 'infer' not in n])
def test_rollout_autograd_matches_reference(name):
    """utils.rollout + the reference's loss + loss.backward() (algorithms/mc_pilco.py:134-197)."""
    import prob_mbrl_amd as pm
    d = common.load(name)
    dyn, pol = common.modules_from_fixture(d, name, DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    H = int(d['H'])
    G = int(d['mm_groups'])
    kw = {}
    if bool(d['mm_states']):
        kw = dict(mm_states=True, mm_rewards=True, mm_groups=G if G > 0 else None,
                  z_mm=torch.tensor(d['z_mm'], device=DEV), z_rr=torch.tensor(d['z_rr'], device=DEV))
    states, actions, rewards = pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False,
                                                resample_action_noise=False, **kw)
    assert len(states) == H + 1 and len(actions) == H and len(rewards) == H
    assert states[0].shape == x0.shape and rewards[0].shape == (x0.shape[0], 1)
    gamma = [float(g) for g in d['gamma']]
    disc = torch.stack([r * gamma[i] for i, r in enumerate(rewards)])
    returns = -disc.sum(0) if bool(d['maximize']) else disc.sum(0)
    loss = returns.mean()
    pol.zero_grad()
    loss.backward()
    assert common.rel(torch.stack(states).detach().cpu().numpy(), d['ref64_states']) < 2e-5
    assert abs(float(loss) - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
    assert common.rel(_flat_grad(pol), d['ref64_grad']) < 1e-4
    # frozen buffers were used, not redrawn
    assert torch.equal(pol.model.fc_nonlin.z.cpu(), torch.tensor(d['pol_z']))


@pytest.mark.parametrize('name', common.fixture_names('mcp'))
def test_mc_pilco_matches_reference_iterations(name):
    """Fixtures from the REAL algorithms.mc_pilco (3-4 Adam iterations, fixed x0)."""
    import prob_mbrl_amd as pm
    d = common.load(name)
    dyn, pol = common.modules_from_fixture(d, name, DEV)
    opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
    x0 = torch.tensor(d['x0'], device=DEV)
    losses = []
    G = int(d['mm_groups'])
    disc = None
    gam = np.asarray(d['gamma'])
    if not np.allclose(gam, gam[0]):
        disc = float(gam[1] / gam[0])
    pm.algorithms.mc_pilco(
        x0, dyn, pol, int(d['H']), opt, None, int(d['mcp_n_iters']), mm_states=bool(d['mm_states']),
        mm_rewards=bool(d['mm_rewards']), mm_groups=G if G > 0 else None, maximize=True,
        clip_grad=float(d['mcp_clip']), discount=disc,
        on_iteration=lambda i, loss, *a: losses.append(float(loss)),
        frozen_noise=dict(z_mm=torch.tensor(d['z_mm']), z_rr=torch.tensor(d['z_rr'])))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
    lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
    final = torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()
    assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    # Adam state is visible through the torch optimiser object (drop-in: same opt reused later)
    st = opt.state[lins[0].weight]
    assert int(st['step']) == int(d['mcp_n_iters'])
    m = torch.cat([opt.state[t]['exp_avg'].reshape(-1) for l in lins for t in (l.weight, l.bias)])
    assert np.allclose(m.cpu().numpy(), d['ref32_mcp_exp_avg'], rtol=1e-3, atol=1e-7)


def test_mc_pilco_autograd_path_options():
    """CVaR + regulariser go through the autograd node; runs and moves the parameters."""
    import prob_mbrl_amd as pm
    d = common.load('nomm_d4')
    dyn, pol = common.modules_from_fixture(d, 'nomm_d4', DEV)
    before = pol.model.fc0.weight.detach().clone()
    opt = torch.optim.Adam(pol.parameters(), 1e-3)
    seen = []
    pm.algorithms.mc_pilco(torch.tensor(d['x0'], device=DEV), dyn, pol, 8, opt, None, 3,
                           cvar_eps=0.3, reg_weight=1e-3,
                           on_iteration=lambda i, loss, s, a, r, disc: seen.append(float(loss)))
    assert len(seen) == 3 and all(np.isfinite(seen))
    assert not torch.equal(before, pol.model.fc0.weight.detach())


def test_rollout_default_call_resamples_noise():
    """rollout(x0, dyn, pol, H) with the reference defaults (fresh output noise per step)."""
    import prob_mbrl_amd as pm
    d = common.load('nomm_d5')
    dyn, pol = common.modules_from_fixture(d, 'nomm_d5', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    with torch.no_grad():
        s1, a1, r1 = pm.utils.rollout(x0, dyn, pol, 6)
        s2, a2, r2 = pm.utils.rollout(x0, dyn, pol, 6)
    assert len(s1) == 7 and torch.isfinite(torch.stack(s1)).all()
    assert not torch.equal(s1[-1], s2[-1])            # new noise each call
    assert torch.equal(s1[0], s2[0])
    # moment matching without a PEGASUS buffer draws fresh z per step
    s3, _, r3 = pm.utils.rollout(x0, dyn, pol, 6, mm_states=True, mm_rewards=True)
    assert torch.isfinite(torch.stack(s3)).all() and torch.isfinite(torch.stack(r3)).all()


def test_mc_pilco_failure_resamples_and_skips(capsys):
    """Rank-deficient moment matching (M <= D rows per group): the reference prints the
    traceback, resamples and skips the optimiser step (algorithms/mc_pilco.py:122-131)."""
    import prob_mbrl_amd as pm
    d = common.load('mmg_d4')
    dyn, pol = common.modules_from_fixture(d, 'mmg_d4', DEV)
    before = pol.model.fc1.weight.detach().clone()
    opt = torch.optim.Adam(pol.parameters(), 1e-3)
    calls = []
    pm.algorithms.mc_pilco(torch.tensor(d['x0'], device=DEV), dyn, pol, 8, opt, None, 2,
                           mm_states=True, mm_rewards=True, mm_groups=20,
                           on_iteration=lambda *a: calls.append(1))
    out = capsys.readouterr()
    assert 'RuntimeError' in out.out
    assert calls == [] and torch.equal(before, pol.model.fc1.weight.detach())


def test_checkpoint_roundtrip_and_resample():
    import prob_mbrl_amd as pm
    d = common.load('nomm_d4')
    dyn, pol = common.modules_from_fixture(d, 'nomm_d4', DEV)
    dyn2, pol2 = common.modules_from_fixture(common.load('mmg_d4'), 'mmg_d4', DEV)
    pol2.load(pol.state_dict())
    dyn2.load(dyn.state_dict())
    assert torch.equal(pol2.model.fc1.weight, pol.model.fc1.weight)
    assert torch.equal(dyn2.model.drop0.concrete_noise, dyn.model.drop0.concrete_noise)
    m0 = pol.model.drop0.noise.clone()
    pol.resample(seed=torch.tensor([5]))
    assert pol.model.drop0.noise.shape == m0.shape and not torch.equal(pol.model.drop0.noise, m0)
    keep = float(pol.model.drop0.noise.mean())
    assert 0.8 < keep < 0.97
    dyn.resample(seed=torch.tensor([5]))
    cn = dyn.model.drop0.concrete_noise
    assert set(np.unique(cn.detach().cpu().numpy())) <= {0.0, 1.0}
    x0 = torch.tensor(d['x0'], device=DEV)
    s, a, r = pm.utils.rollout(x0, dyn, pol, 4, resample_state_noise=False,
                               resample_action_noise=False)
    assert torch.isfinite(torch.stack(s)).all()


@pytest.mark.gpu
@pytest.mark.parametrize('name', common.fixture_names('standalone'))
def test_standalone_forwards_match_reference(name):
    """Policy.forward / Regressor.forward / DynamicsModel.forward outside a rollout
    (models/core.py:169-187, 221-248, 265-303) through pmbrl_mlp_forward."""
    d = common.load(name)
    dyn, pol = common.modules_from_fixture(d, name)
    dev = torch.device('cuda:0')
    x = torch.tensor(d['x0'], dtype=torch.float32, device=dev)
    tol = dict(rtol=3e-5, atol=3e-6)
    a = pol(x, resample=False, return_samples=True, resample_noise=False)
    assert a.shape == d['ref32_act'].shape and a.is_cuda
    assert np.allclose(a.cpu().numpy(), d['ref64_act'], **tol)
    a_ns = pol(x, resample=False, return_samples=False)
    assert np.allclose(a_ns.cpu().numpy(), d['ref64_act_nosample'], **tol)
    mean, log_std = dyn((x, a), resample=False)
    assert np.allclose(mean.cpu().numpy(), d['ref64_dyn_mean'], **tol)
    assert np.allclose(log_std.cpu().numpy(), d['ref64_dyn_log_std'], **tol)
    nxt, rew = dyn((x, a), return_samples=True, separate_outputs=True, deltas=False, resample=False,
                   resample_noise=False)
    assert np.allclose(nxt.cpu().numpy(), d['ref64_next'], **tol)
    assert np.allclose(rew.cpu().numpy().reshape(-1, 1), d['ref64_rew'], **tol)
    dlt, _ = dyn((x, a), return_samples=True, separate_outputs=True, deltas=True, resample=False,
                 resample_noise=False)
    assert np.allclose(dlt.cpu().numpy(), d['ref64_delta'], rtol=1e-4, atol=1e-6)
    # numpy in -> numpy out, a single state -> one row, fewer rows than the stored mask reuse its first rows
    a1 = pol(np.asarray(d['x0'][0], dtype=np.float32), resample=False, resample_noise=False)
    assert isinstance(a1, np.ndarray) and a1.shape == (1, a.shape[1])
    a5 = pol(x[:5], resample=False, resample_noise=True)
    assert a5.shape == (5, a.shape[1]) and bool(torch.isfinite(a5).all())
    # resample=True: fresh masks -> different but bounded actions (|a| <= scale + |bias|)
    b1 = pol(x, resample=True, resample_noise=True)
    b2 = pol(x, resample=True, resample_noise=True)
    assert not torch.equal(b1, b2)
    bound = (pol.scale.abs() + pol.bias.abs()).to(dev) + 1e-5
    assert bool((b1.abs() <= bound).all())


@pytest.mark.gpu
def test_train_regressor_fits_a_dataset():
    """utils.train_regressor end to end on the device: the data log-likelihood improves, the
    optimiser state stays usable, the trained model evaluates through the stand-alone forward."""
    import prob_mbrl_amd as pm
    torch.manual_seed(0)
    np.random.seed(0)
    dev = torch.device('cuda:0')
    D, U, N = 3, 1, 256
    dyn = pm.models.DynamicsModel(
        pm.models.mlp(D + U, 2 * D, [64, 64],
                      dropout_layers=[pm.models.CDropout(0.1 * np.ones(64)) for _ in range(2)],
                      nonlin=torch.nn.ReLU),
        reward_func=pm.rewards.RendezvousReward(Q=torch.eye(3), R=torch.eye(1)) if False else None,
        output_density=pm.models.DiagGaussianDensity(D)).float().to(dev)
    X = torch.randn(N, D + U, device=dev)
    Y = torch.tanh(X[:, :D] + 0.5 * X[:, D:D + 1]) + 0.05 * torch.randn(N, D, device=dev)
    dyn.set_dataset(X, Y)
    opt = torch.optim.Adam(dyn.parameters(), 2e-3)
    l0 = pm.utils.train_regressor(dyn, iters=5, batchsize=64, optimizer=opt).clone()
    l1 = pm.utils.train_regressor(dyn, iters=600, batchsize=64, optimizer=opt).clone()
    assert bool(torch.isfinite(l1).all())
    assert float(l1[1]) < float(l0[1]) - 0.5, (l0, l1)        # E[-lml] went down
    st = opt.state[next(iter(dyn.parameters()))]
    assert int(st['step']) == 6 + 601 and st['exp_avg'].shape == next(iter(dyn.parameters())).shape
    assert not dyn.training
    mean, log_std = dyn(X[:32], resample=False)
    err = (mean - Y[:32]).abs().mean()
    assert float(err) < 0.35 and bool(torch.isfinite(log_std).all())
