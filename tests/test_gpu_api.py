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


def test_rollout_resamples_masks_every_step(monkeypatch):
    """rollout(resample_model=True, resample_policy=True): new dropout masks at every step
    (utils/rollout.py:95-98,110-115).  Fixture from the reference's own rollout, whose draws were
    recorded; here they are handed out by the dropout modules' mask source in the same order."""
    import prob_mbrl_amd as pm
    d = common.load('stepmask_d4')
    H = int(d['H'])
    frozen = dict(d)
    for k in ('pol_mask0', 'pol_mask1', 'dyn_mask0', 'dyn_mask1'):
        frozen[k] = d[k][0]
    dyn, pol = common.modules_from_fixture(frozen, 'stepmask_d4', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    queues = {id(pol.model.drop0): list(d['pol_mask0']), id(pol.model.drop1): list(d['pol_mask1']),
              id(dyn.model.drop0): list(d['dyn_mask0']), id(dyn.model.drop1): list(d['dyn_mask1'])}

    def fake(self, H_, B, width):
        # the recorded draws of all H steps, packed the way the device draw (pmbrl_draw_masks) hands them over
        from prob_mbrl_amd import engine as E
        ms = torch.stack([torch.tensor(queues[id(self)].pop(0), device=DEV) for _ in range(H_)])
        return E.pack_mask(ms.reshape(H_ * B, width).float().contiguous())

    monkeypatch.setattr(pm.models.BDropout, 'step_mask_bits', fake)
    monkeypatch.setattr(pm.models.CDropout, 'step_mask_bits', fake)
    states, actions, rewards = pm.utils.rollout(x0, dyn, pol, H, resample_model=True, resample_policy=True,
                                                resample_state_noise=False, resample_action_noise=False)
    assert all(len(q) == 0 for q in queues.values())
    gamma = [float(g) for g in d['gamma']]
    loss = (-torch.stack([r * gamma[i] for i, r in enumerate(rewards)]).sum(0)).mean()
    pol.zero_grad()
    loss.backward()
    assert common.rel(torch.stack(states).detach().cpu().numpy(), d['ref64_states']) < 2e-5
    assert abs(float(loss) - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
    assert common.rel(_flat_grad(pol), d['ref64_grad']) < 1e-4
    monkeypatch.undo()
    # and with the modules' own draws: runs, finite, different masks at different steps
    s2, _, _ = pm.utils.rollout(x0, dyn, pol, H, resample_model=True, resample_policy=True)
    assert torch.isfinite(torch.stack(s2)).all()


def test_mc_pilco_with_per_unit_dropout_rates_steps():
    """mc_pilco on a policy with per-unit dropout rates: the device-side Adam step is not taken (it would update the
    scaled copies' originals with the wrong gradient); the autograd form runs, the loss is finite, parameters move."""
    import prob_mbrl_amd as pm
    d = common.load('unit_rates_d4')
    dyn, pol = common.modules_from_fixture(d, 'unit_rates_d4', DEV)
    opt = torch.optim.Adam(pol.parameters(), 1e-3)
    before = torch.cat([p.detach().reshape(-1) for p in pol.parameters()]).clone()
    x0 = torch.tensor(d['x0'], device=DEV)
    pm.algorithms.mc_pilco(x0, dyn, pol, int(d['H']), opt, opt_iters=3, pegasus=True)
    after = torch.cat([p.detach().reshape(-1) for p in pol.parameters()])
    assert torch.isfinite(after).all() and not torch.equal(after, before)


def test_standalone_policy_with_per_unit_rates_is_the_policy_the_rollout_runs():
    """Per-unit BDropout rates (models/modules.py:19-27,55-61: (x * mask) / p with p a vector): the stand-alone
    evaluation behind Policy.__call__ / apply_controller must apply the same 1 / p_j the rollout folds into the layer
    in front of the dropout.  The first action of a rollout IS policy(x0) with the stored masks and frozen noise."""
    import prob_mbrl_amd as pm
    d = common.load('unit_rates_d4')
    dyn, pol = common.modules_from_fixture(d, 'unit_rates_d4', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    _, actions, _ = pm.utils.rollout(x0, dyn, pol, int(d['H']), resample_state_noise=False,
                                     resample_action_noise=False)
    u = pol(x0, resample=False, resample_noise=False)
    a0 = actions[0].detach()
    assert common.rel(u.detach().cpu().numpy(), a0.cpu().numpy()) < 2e-6
    # ... and differs from what dropping the 1 / p_j would give (the rates of the fixture are not all equal)
    rates = pol.model.drop0.rate
    assert rates.numel() > 1 and float(rates.max() - rates.min()) > 0


@pytest.mark.parametrize('name', [n for n in common.fixture_names('iter') if not n.startswith('stepmask')] +
                         ['unit_rates_d4', 'unit_rates_d4_mmg'])
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
                  z_mm=torch.tensor(d['z_mm'], device=DEV), z_rr=torch.tensor(d['z_rr'], device=DEV),
                  infer_noise_variables=bool(d['infer_ns']) if 'infer_ns' in d else False)
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
    from prob_mbrl_amd import rollout as RO
    served_before = {k: e.reg_calls() for k, e in RO._ENGINES.items()}
    pm.algorithms.mc_pilco(
        x0, dyn, pol, int(d['H']), opt, None, int(d['mcp_n_iters']), mm_states=bool(d['mm_states']),
        mm_rewards=bool(d['mm_rewards']), mm_groups=G if G > 0 else None, maximize=True,
        clip_grad=float(d['mcp_clip']), discount=disc,
        cvar_eps=float(d['mcp_cvar_eps']) if 'mcp_cvar_eps' in d else 0.0,       # algorithms/mc_pilco.py:146-154
        reg_weight=float(d['mcp_reg_weight']) if 'mcp_reg_weight' in d else 0.0,  # algorithms/mc_pilco.py:193-194
        on_iteration=lambda i, loss, *a: losses.append(float(loss)),
        frozen_noise=dict(z_mm=torch.tensor(d['z_mm']), z_rr=torch.tensor(d['z_rr'])))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
    if name in ('mcp_full200', 'mcp_full200_mmg'):
        # the 2 x 200 shape of BASELINE.json's metric: every iteration's sweeps ran on the register-resident family
        # (csrc/pmbrl_reg.h) -- the engines mc_pilco builds are cached per shape (rollout._ENGINES)
        n = int(d['mcp_n_iters'])
        deltas = [tuple(a - b for a, b in zip(e.reg_calls(), served_before.get(k, (0, 0))))
                  for k, e in RO._ENGINES.items() if e.info['reg']]
        assert (n, n) in deltas, deltas
    lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
    final = torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()
    # Adam divides by |g_i|: an element whose gradient is tiny against the gradient's rms moves by up to lr per step on
    # the SIGN of rounding noise -- in any fp32-class arithmetic, the reference's own included.  The bar per element is
    # the north star's 1e-4 (of the gradient's rms) pushed through Adam's normalisation: n lr eps_g / (|g_i| + eps),
    # capped at n lr; |g_i| from the reference's second-moment estimate.  (The 32-unit fixtures never need the term:
    # 2e-6 covers them; of the 41 602 parameters of the 2 x 200 policy a few dozen have |g_i| < 1e-3 rms.)
    n_it, lr = int(d['mcp_n_iters']), float(d['mcp_lr'])
    g_abs = np.sqrt(np.asarray(d['ref32_mcp_exp_avg_sq'], np.float64) / (1.0 - 0.999 ** n_it))
    eps_g = 1e-4 * np.sqrt(np.mean(g_abs ** 2))
    tol = 1e-4 * np.abs(d['ref32_mcp_final']) + 2e-6 + n_it * lr * np.minimum(1.0, eps_g / (g_abs + 1e-8))
    err = np.abs(final.astype(np.float64) - d['ref32_mcp_final'])
    plain = 1e-4 * np.abs(d['ref32_mcp_final']) + 2e-6
    print('%s: parameters beyond the plain bar %d of %d (worst %.2e at |g| = %.2e, gradient rms %.2e)' %
          (name, int((err > plain).sum()), err.size, float(err.max()), float(g_abs[np.argmax(err)]), float(eps_g * 1e4)))
    assert np.all(err <= tol), (int((err > tol).sum()), float((err - tol).max()), float(err.max()))
    # FROZEN (round 6): these bars may only tighten.  The per-element bar above admits whatever Adam can reach for an element
    # whose gradient is rounding noise; what guards against a systematic error is (a) the COUNT of elements beyond the plain
    # bar -- a regression threshold, measured 0 in all seven fixtures on the round-6 build -- and (b) the update direction,
    # a quantity rounding does not move: the cosine between this run's total parameter update and the reference's.
    n_beyond = int((err > plain).sum())
    assert n_beyond <= (4 if err.size > 40000 else 0), n_beyond
    from prob_mbrl_amd.problem import flat_params
    init = flat_params(d, 'pol').astype(np.float64)          # the policy the fixture's run started from
    upd = final.astype(np.float64) - init
    upd_ref = np.asarray(d['ref32_mcp_final'], np.float64) - init
    cos = float(upd @ upd_ref / (np.linalg.norm(upd) * np.linalg.norm(upd_ref)))
    print('%s: cosine of the total update 1 - %.2e' % (name, 1.0 - cos))
    assert cos > 1.0 - 1e-7, cos          # (measured: 1 - 6.3e-9 at worst, mcp_mm1)
    # Adam state is visible through the torch optimiser object (drop-in: same opt reused later)
    st = opt.state[lins[0].weight]
    assert int(st['step']) == int(d['mcp_n_iters'])
    m = torch.cat([opt.state[t]['exp_avg'].reshape(-1) for l in lins for t in (l.weight, l.bias)])
    # the first moment against the reference's, on the moment's own scale: 1e-3 of the element + 1e-4 of the vector's rms
    # (north_star's bar for the gradient this is the running mean of).  Until round 5 the floor was a fixed 1e-7 -- the size
    # of the whole vector in three of the fixtures (rms 1e-7 .. 2e-6: nothing was compared) and 4e-5 of the rms in mcp_mm1,
    # the split-precision gradient's own noise (one element of 1314 sat 4e-9 beyond it)
    m_ref = np.asarray(d['ref32_mcp_exp_avg'], dtype=np.float64)
    m_err = np.abs(m.cpu().numpy().astype(np.float64) - m_ref)
    m_tol = 1e-3 * np.abs(m_ref) + 1e-4 * np.sqrt(np.mean(m_ref ** 2))
    assert np.all(m_err <= m_tol), (int((m_err > m_tol).sum()), float((m_err - m_tol).max()))
    # ... and as a vector (rounding-insensitive: the first moment is a running mean of the gradients, linear in them):
    # north_star's bar for the gradient itself
    m_rel = float(np.linalg.norm(m.cpu().numpy().astype(np.float64) - m_ref) / np.linalg.norm(m_ref))
    print('%s: first moment, relative L2 error %.2e' % (name, m_rel))
    assert m_rel < 5e-5, m_rel           # (measured: 1.4e-5 at worst, mcp_mm1)


def test_rollout_truncated_horizon_matches_reference(monkeypatch):
    """rollout() after a failure in step 8 of 12 (> 5 steps done): the caller gets the first 8 steps
    and optimises on them, like the reference (utils/rollout.py:154-157); fixture from the
    reference's own rollout.  The failure is injected where the host learns about it (the status
    word read by Engine.valid_steps); the trajectory tensors behind step 8 are poisoned."""
    import prob_mbrl_amd as pm
    from prob_mbrl_amd import engine as E
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    dyn, pol = common.modules_from_fixture(d, 'trunc_mm', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    H = int(d['H'])
    real = E.Engine.valid_steps

    def failing(self):
        assert real(self) == H
        self.status[0] = n
        S, A, R = self._traj
        S[n + 1:], A[n:], R[n:] = float('nan'), float('nan'), float('nan')
        return n

    monkeypatch.setattr(E.Engine, 'valid_steps', failing)
    states, actions, rewards = pm.utils.rollout(
        x0, dyn, pol, H, resample_state_noise=False, resample_action_noise=False, mm_states=True,
        mm_rewards=True, z_mm=torch.tensor(d['z_mm'], device=DEV), z_rr=torch.tensor(d['z_rr'], device=DEV))
    assert len(states) == n + 1 and len(actions) == n and len(rewards) == n
    gamma = [float(g) for g in d['gamma']]
    loss = (-torch.stack([r * gamma[i] for i, r in enumerate(rewards)]).sum(0)).mean()
    pol.zero_grad()
    loss.backward()
    assert common.rel(torch.stack(states).detach().cpu().numpy(), d['ref64_states']) < 2e-5
    assert abs(float(loss) - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
    g = _flat_grad(pol)
    assert np.all(np.isfinite(g)) and common.rel(g, d['ref64_grad']) < 1e-4
    # five steps or fewer: the reference re-raises (utils/rollout.py:155-157)
    monkeypatch.setattr(E.Engine, 'valid_steps', lambda self: 5)
    with pytest.raises(RuntimeError):
        pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False, resample_action_noise=False)


def test_fp16_range_failure_is_retried_on_bf16_pieces(capsys):
    """'split_f16' (the default arithmetic) keeps the forward sweep's hidden activations as fp16 pieces:
    beyond +-65504 they overflow and the step is reported as non-finite.  That must not look like a
    failed rollout to the caller: rollout() and mc_pilco re-run on the bf16 pieces (fp32's range) before
    believing a failure.  Policy rescaled so that its first hidden layer is ~1e6 x larger and the second
    layer's weights 1e-6 x smaller: the same function in exact arithmetic, far outside fp16."""
    import prob_mbrl_amd as pm
    from prob_mbrl_amd import engine as E
    assert E.get_precision() == 'split_f16'
    d = dict(common.load('nomm_d4'))
    d['pol_W0'] = d['pol_W0'] * np.float32(2.0**20)
    d['pol_b0'] = d['pol_b0'] * np.float32(2.0**20)
    d['pol_W1'] = d['pol_W1'] * np.float32(2.0**-20)       # exact rescaling (powers of two)
    ref = common.load('nomm_d4')
    dyn, pol = common.modules_from_fixture(d, 'nomm_d4', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    H = int(d['H'])
    # the fp16 path alone fails ...
    eng, args, _ = common.engine_from_fixture(d, DEV, precision='split_f16')
    eng.forward(**args)
    assert eng.valid_steps() < H
    # ... the public entry points do not
    states, actions, rewards = pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False,
                                                resample_action_noise=False)
    assert len(rewards) == H
    assert common.rel(torch.stack(states).detach().cpu().numpy(), ref['ref64_states']) < 2e-5
    opt = torch.optim.Adam(pol.parameters(), 1e-4)
    seen = []
    pm.algorithms.mc_pilco(x0, dyn, pol, H, opt, None, 6, on_iteration=lambda i, loss, *a: seen.append(float(loss)),
                           frozen_noise=dict(z_mm=torch.zeros(H + x0.shape[0], 4), z_rr=torch.zeros(H + x0.shape[0], 1)))
    out = capsys.readouterr().out
    assert 'continuing with split' in out and 'RuntimeError' not in out
    assert len(seen) >= 4 and all(np.isfinite(seen))
    assert abs(seen[0] - float(ref['ref32_loss'])) <= 1e-4 * abs(float(ref['ref32_loss']))


def test_fp16_range_failure_in_the_general_family_is_retried_in_fp32():
    """The same hazard in the general kernel family (here: angle_dims inside the networks, which only that family
    runs): its split form has fp16 pieces only, so the retry lands on exact fp32."""
    import prob_mbrl_amd as pm
    name = 'angles_d4'
    d = dict(common.load(name))
    d['pol_W0'] = d['pol_W0'] * np.float32(2.0**20)
    d['pol_b0'] = d['pol_b0'] * np.float32(2.0**20)
    d['pol_W1'] = d['pol_W1'] * np.float32(2.0**-20)
    ref = common.load(name)
    eng, args, _ = common.engine_from_fixture(d, DEV, precision='split_f16')
    assert not eng.info['fast'] and eng.info['precision'] == 'split_f16'
    eng.forward(**args)
    assert eng.valid_steps() < int(d['H'])
    eng, args, _ = common.engine_from_fixture(d, DEV, precision='split')      # bf16 pieces: not in this family
    assert eng.info['precision'] == 'f32'
    dyn, pol = common.modules_from_fixture(d, name, DEV)
    states, actions, rewards = pm.utils.rollout(torch.tensor(d['x0'], device=DEV), dyn, pol, int(d['H']),
                                                resample_state_noise=False, resample_action_noise=False)
    assert len(rewards) == int(d['H'])
    assert common.rel(torch.stack(states).detach().cpu().numpy(), ref['ref64_states']) < 2e-5


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


def test_mixture_head_trains_and_rolls_out():
    """A dynamics model with a GaussianMixtureDensity head (examples/deep_pilco_mm.py:117-121, --dyn_components > 1)
    through the public API: utils.train_regressor with the mixture log-likelihood (loss goes down), then
    utils.rollout with it and a policy gradient out of loss.backward()."""
    from functools import partial

    import prob_mbrl_amd as pm
    torch.manual_seed(0)
    np.random.seed(0)
    dev = torch.device('cuda:0')
    D, U, N, n = 4, 1, 256, 3
    dyn = pm.models.DynamicsModel(
        pm.models.mlp(D + U, (2 * D + 1) * n + 1, [64, 64],
                      dropout_layers=[pm.models.CDropout(0.1 * np.ones(64)) for _ in range(2)], nonlin=torch.nn.ReLU),
        reward_func=pm.rewards.CartpoleReward(pole_length=torch.tensor(0.5)),
        output_density=pm.models.GaussianMixtureDensity(D, n)).float().to(dev)
    X = torch.randn(N, D + U, device=dev)
    # two-mode targets: what a single Gaussian cannot fit
    sign = torch.where(torch.rand(N, 1, device=dev) < 0.5, -1.0, 1.0)
    Y = 0.05 * X[:, :D] + 0.1 * sign * torch.ones(1, D, device=dev) + 0.01 * torch.randn(N, D, device=dev)
    dyn.set_dataset(X, Y)
    opt = torch.optim.Adam(dyn.parameters(), 2e-3)
    ll = pm.losses.gaussian_mixture_log_likelihood
    with pytest.raises(NotImplementedError):
        pm.utils.train_regressor(dyn, iters=1, batchsize=64, optimizer=opt)      # Gaussian likelihood, mixture head
    l0 = pm.utils.train_regressor(dyn, iters=5, batchsize=64, optimizer=opt, log_likelihood=ll).clone()
    l1 = pm.utils.train_regressor(dyn, iters=500, batchsize=64, optimizer=opt, log_likelihood=ll).clone()
    assert bool(torch.isfinite(l1).all()) and float(l1[1]) < float(l0[1]) - 0.5, (l0, l1)
    pol = pm.models.Policy(
        pm.models.mlp(D, 2 * U, [32, 32], dropout_layers=[pm.models.BDropout(0.1) for _ in range(2)],
                      nonlin=torch.nn.ReLU, output_nonlin=partial(pm.models.DiagGaussianDensity, U)),
        np.array([10.0], dtype=np.float32)).float().to(dev)
    dyn.eval()
    x0 = 0.1 * torch.randn(50, D, device=dev)
    H = 6
    states, actions, rewards = pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False,
                                                resample_action_noise=False)
    assert len(states) == H + 1 and bool(torch.isfinite(torch.stack(states)).all())
    zpi = dyn.output_density.z_pi.clone()
    loss = -torch.stack(rewards).sum(0).mean()
    pol.zero_grad()
    loss.backward()
    g = _flat_grad(pol)
    assert np.isfinite(g).all() and np.abs(g).max() > 0
    # the Gumbel noise is frozen across rollouts, the component draws and the Gaussian noise are not
    s2, _, _ = pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False, resample_action_noise=False)
    assert torch.equal(dyn.output_density.z_pi, zpi) and not torch.equal(torch.stack(s2), torch.stack(states))
    # stand-alone evaluation (models/core.py:169-187): distribution parameters, and their likelihood of the data
    mean, log_std, logit_pi = dyn(X[:64], resample=False)
    assert mean.shape == (64, D, n) and log_std.shape == (64, D, n) and logit_pi.shape == (64, n)
    lp = dyn.output_density.log_prob(Y[:64], mean, log_std, logit_pi)
    assert lp.shape == (64, 1) and float(lp.mean()) > 2.0       # both modes found: far above one broad Gaussian
    samples = dyn((X[:64, :D], X[:64, D:]), return_samples=True, separate_outputs=True, resample=False)[0]
    assert samples.shape == (64, D) and bool(torch.isfinite(samples).all())


def _value_from_fixture(d):
    """The critic of examples/deep_pilco_no_mm_with_value.py:269-278 (no output density,
    concrete dropout, eval mode) holding the fixture's weights and masks."""
    import prob_mbrl_amd as pm
    n = int(d['val_n_layers'])
    hid = [d['val_W%d' % i].shape[0] for i in range(n - 1)]
    D = d['val_W0'].shape[1]
    V = pm.models.Regressor(pm.models.mlp(
        D, 1, hid, dropout_layers=[pm.models.CDropout(0.1 * np.ones(h)) for h in hid],
        nonlin=torch.nn.ReLU)).float()
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    with torch.no_grad():
        lins = [m for m in V.model._modules.values() if isinstance(m, torch.nn.Linear)]
        for i, lin in enumerate(lins):
            lin.weight.copy_(T(d['val_W%d' % i]))
            lin.bias.copy_(T(d['val_b%d' % i]))
        for i in range(n - 1):
            dr = getattr(V.model, 'drop%d' % i)
            dr.noise.data = torch.rand(d['val_mask%d' % i].shape)
            dr.concrete_noise = T(d['val_mask%d' % i])
        for k in ('mx', 'iSx', 'my', 'Sy'):
            getattr(V, k).data = T(d['val_' + k]).reshape(1, -1)
    return V.to(DEV).eval()


def _flat_params(pol):
    lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
    return torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()


def test_mc_pilco_value_bootstrap_matches_reference():
    """mc_pilco(value_func=V): fixture from the reference's own run (algorithms/mc_pilco.py:136-140).
    dV/ds_H comes from pmbrl_mlp_grad_input and enters the adjoint sweep as grad_states[H]."""
    import prob_mbrl_amd as pm
    d = common.load('ext_value')
    dyn, pol = common.modules_from_fixture(d, 'ext_value', DEV)
    V = _value_from_fixture(d)
    x0 = torch.tensor(d['x0'], device=DEV)
    # the critic alone: value and input gradient against the oracle
    from oracle import ref_torch as R
    val = R.value_from_npz(d, torch.float64)
    xr = torch.tensor(d['x0'], dtype=torch.float64, requires_grad=True)
    vr = R.value_forward(xr, val)
    vr.sum().backward()
    xd = x0.clone().requires_grad_(True)
    vd = V(xd, resample=False, return_samples=True)
    assert vd.shape == (x0.shape[0], 1)
    vd.sum().backward()
    assert common.rel(vd.detach().cpu().numpy(), vr.detach().numpy()) < 2e-6
    assert common.rel(xd.grad.cpu().numpy(), xr.grad.numpy()) < 1e-5
    opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
    losses = []
    pm.algorithms.mc_pilco(
        x0, dyn, pol, int(d['H']), opt, None, int(d['mcp_n_iters']), value_func=V, maximize=True,
        clip_grad=float(d['mcp_clip']), on_iteration=lambda i, loss, *a: losses.append(float(loss)),
        frozen_noise=dict(z_mm=torch.zeros(1, 4), z_rr=torch.zeros(1, 1)))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
    assert np.allclose(_flat_params(pol), d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)


def test_mc_pilco_prioritized_replay_matches_reference():
    """mc_pilco(prioritized_replay=True) over an ExperienceDataset: same sampled start states in
    every iteration, same losses / parameters / final priorities as the reference's run
    (algorithms/mc_pilco.py:80-84, 156-188, 222-246).  The per-step ||dL/da_t|| come out of the
    adjoint sweep (pmbrl_rollout_bwd's action_grad_norms) instead of tensor hooks."""
    import prob_mbrl_amd as pm
    from prob_mbrl_amd import algorithms as ALG
    d = common.load('ext_replay')
    dyn, pol = common.modules_from_fixture(d, 'ext_replay', DEV)
    exp = pm.utils.ExperienceDataset()
    for e in range(int(d['replay_n_episodes'])):
        st = d['replay_states%d' % e]
        T = len(st)
        exp.append_episode(list(st), list(np.zeros((T, 1), np.float32)), list(np.zeros(T)), [None] * T, None)
    ALG.x0_tree, ALG.episode_counter = None, 0
    np.random.seed(int(d['replay_np_seed']))
    opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
    losses, x0s = [], []
    pm.algorithms.mc_pilco(
        torch.tensor(d['x0'], device=DEV), dyn, pol, int(d['H']), opt, exp, int(d['mcp_n_iters']),
        maximize=True, clip_grad=float(d['mcp_clip']), prioritized_replay=True, priority_alpha=0.6,
        init_priority_beta=0.4, priority_beta_increase=0.1,
        on_rollout=lambda i, s, a, r, disc: x0s.append(s[0].detach().cpu().numpy()),
        on_iteration=lambda i, loss, *a: losses.append(float(loss)),
        frozen_noise=dict(z_mm=torch.zeros(1, 4), z_rr=torch.zeros(1, 1)))
    assert np.array_equal(np.stack(x0s), d['replay_x0s'].astype(np.float32))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=1e-4)
    assert np.allclose(_flat_params(pol), d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    tree = ALG.x0_tree
    n = len(d['replay_final_counts'])
    assert tree.size == n and np.array_equal(tree.counts[:n], d['replay_final_counts'])
    assert np.allclose(tree.sum_tree[tree.max_size - 1:tree.max_size - 1 + n],
                       d['replay_final_leaves'], rtol=1e-3)
    ALG.x0_tree, ALG.episode_counter = None, 0


def test_example_script_end_to_end(tmp_path):
    """examples/deep_pilco.py = the loop of the reference's example scripts (apply_controller ->
    ExperienceDataset -> train_regressor -> mc_pilco, checkpoints on disk) on the self-contained
    cart-pole: two short policy-search rounds, with and without moment matching."""
    import importlib.util
    import os
    import prob_mbrl_amd as pm
    spec = importlib.util.spec_from_file_location('deep_pilco_example',
                                                  os.path.join(common.ROOT, 'examples', 'deep_pilco.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    for extra in ([], ['--no_mm']):
        folder, hist = mod.main(['-o', str(tmp_path), '--ps_iters', '2', '--control_H', '25', '--pred_H', '10',
                                 '--dyn_opt_iters', '150', '--pol_opt_iters', '25', '--pol_batch_size', '50',
                                 '--dyn_shape', '64,64', '--pol_shape', '64,64', '--n_initial_epi', '2'] + extra)
        assert len(hist) == 2 and all(np.isfinite(h['last_loss']) for h in hist)
        assert hist[-1]['n_samples'] == 4 * 25
        for f in ('experience.pth.tar', 'latest_dynamics.pth.tar', 'latest_policy.pth.tar', 'args.pth.tar'):
            assert os.path.exists(os.path.join(folder, f))
        # the checkpoint loads back into fresh modules (utils.load_checkpoint)
        env = pm.envs.Cartpole()
        dyn = pm.models.DynamicsModel(
            pm.models.mlp(5, 8, [64, 64], dropout_layers=[pm.models.CDropout(0.1 * np.ones(64)) for _ in range(2)],
                          nonlin=torch.nn.ReLU),
            reward_func=env.reward_func, output_density=pm.models.DiagGaussianDensity(4)).float()
        from functools import partial
        pol = pm.models.Policy(
            pm.models.mlp(4, 2, [64, 64], dropout_layers=[pm.models.BDropout(0.1) for _ in range(2)],
                          nonlin=torch.nn.ReLU, output_nonlin=partial(pm.models.DiagGaussianDensity, 1)),
            env.action_space.high, env.action_space.low).float()
        exp = pm.utils.ExperienceDataset()
        pm.utils.load_checkpoint(folder, dyn, pol, exp)
        assert exp.n_samples() == 100 and exp.n_episodes() == 4
        sd = torch.load(os.path.join(folder, 'latest_policy.pth.tar'), weights_only=False)
        assert torch.equal(pol.model.fc0.weight.cpu(), sd['model.fc0.weight'].cpu())


def test_critic_fit_matches_reference():
    """prob_mbrl_amd.critic.update_value_function against the reference example's own function
    (fixture: three updates with the recorded Bernoulli outcomes): one pmbrl_bnn_loss_grad call
    with the MSE head per update."""
    import prob_mbrl_amd as pm
    from prob_mbrl_amd.critic import update_value_function
    d = common.load('critic_fit')
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    n = int(d['n_layers'])
    hid = [d['W%d_init' % i].shape[0] for i in range(n - 1)]
    D = d['W0_init'].shape[1]
    V = pm.models.Regressor(pm.models.mlp(
        D, 1, hid, dropout_layers=[pm.models.CDropout(0.2 * np.ones(h)) for h in hid], nonlin=torch.nn.ReLU)).float()
    with torch.no_grad():
        lins = [m for m in V.model._modules.values() if isinstance(m, torch.nn.Linear)]
        for i, lin in enumerate(lins):
            lin.weight.copy_(T(d['W%d_init' % i]))
            lin.bias.copy_(T(d['b%d_init' % i]))
        for i in range(n - 1):
            dr = getattr(V.model, 'drop%d' % i)
            dr.logit_p.copy_(T(d['logit_p%d_init' % i]))
            dr.noise.data = T(d['u%d' % i])
            dr.concrete_noise = torch.ones_like(dr.noise)
        for k in ('mx', 'iSx', 'my', 'Sy'):
            getattr(V, k).data = T(d[k]).reshape(1, -1)
    V = V.to(DEV)
    opt = torch.optim.Adam(V.parameters(), float(d['lr']))
    H, gam = int(d['H']), float(d['gamma'])
    states = [None] * (H + 1)
    states[0], states[H] = T(d['states0']).to(DEV), T(d['statesH']).to(DEV)
    rewards = [T(r).to(DEV) for r in d['rewards']]
    for it in range(int(d['n_updates'])):
        draws = [d['hard%d_it%d' % (k, it)] for k in range(2 * (n - 1))]
        update_value_function(V, opt, H, it, states, None, rewards, lambda i: gam**i,
                              reg_weight=float(d['reg_weight']), _bernoulli=draws)
        for i, lin in enumerate(lins):
            assert np.allclose(lin.weight.detach().cpu().numpy(), d['W%d_it%d' % (i, it)], rtol=5e-4, atol=5e-6), (it, i)
            assert np.allclose(lin.bias.detach().cpu().numpy(), d['b%d_it%d' % (i, it)], rtol=5e-4, atol=5e-6)
        for i in range(n - 1):
            assert np.allclose(getattr(V.model, 'drop%d' % i).logit_p.detach().cpu().numpy(),
                               d['logit_p%d_it%d' % (i, it)], rtol=5e-4, atol=5e-6)
    assert not V.training


def test_mc_pilco_with_value_function_and_critic_hook():
    """The third example's arrangement (examples/deep_pilco_no_mm_with_value.py:380-400): mc_pilco with
    value_func=V and on_rollout=update_value_function fitting V between rollouts."""
    from functools import partial
    import prob_mbrl_amd as pm
    from prob_mbrl_amd.critic import update_value_function
    d = common.load('ext_value')
    dyn, pol = common.modules_from_fixture(d, 'ext_value', DEV)
    V = _value_from_fixture(d)
    optV = torch.optim.Adam(V.parameters(), 1e-3)
    opt = torch.optim.Adam(pol.parameters(), 1e-3)
    H = int(d['H'])
    v_before = V.model.fc0.weight.detach().clone()
    p_before = pol.model.fc0.weight.detach().clone()
    losses = []
    pm.algorithms.mc_pilco(torch.tensor(d['x0'], device=DEV), dyn, pol, H, opt, None, 4, value_func=V,
                           on_rollout=partial(update_value_function, V, optV, H),
                           on_iteration=lambda i, loss, *a: losses.append(float(loss)))
    assert len(losses) == 4 and all(np.isfinite(losses))
    assert not torch.equal(v_before, V.model.fc0.weight.detach())
    assert not torch.equal(p_before, pol.model.fc0.weight.detach())
    assert torch.isfinite(V.model.fc0.weight).all() and int(optV.state[V.model.fc0.weight]['step']) == 4


@pytest.mark.parametrize('mode', ['decoupled', 'prioritized'])
def test_train_regressor_options_match_reference(mode):
    """utils.train_regressor(decoupled_reg=True) / (prioritized_sampling=True): fixtures from the
    reference's own function (10 steps, recorded concrete-dropout draws, numpy seeded)."""
    import prob_mbrl_amd as pm
    from prob_mbrl_amd import train_regressor as TR
    d = common.load('bnnopt_' + mode)
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    n = int(d['n_layers'])
    hid = [d['W%d_init' % i].shape[0] for i in range(n - 1)]
    Din, Dout = d['W0_init'].shape[1], d['W%d_init' % (n - 1)].shape[0] // 2
    dyn = pm.models.DynamicsModel(
        pm.models.mlp(Din, 2 * Dout, hid, dropout_layers=[pm.models.CDropout(0.25 * np.ones(h)) for h in hid],
                      nonlin=torch.nn.ReLU),
        reward_func=None, output_density=pm.models.DiagGaussianDensity(Dout)).float()
    lins = [m for m in dyn.model._modules.values() if isinstance(m, torch.nn.Linear)]
    with torch.no_grad():
        for i, lin in enumerate(lins):
            lin.weight.copy_(T(d['W%d_init' % i]))
            lin.bias.copy_(T(d['b%d_init' % i]))
        for i in range(n - 1):
            getattr(dyn.model, 'drop%d' % i).logit_p.copy_(T(d['logit_p%d_init' % i]))
    dyn.set_dataset(T(d['X']), T(d['Y']))
    opt = torch.optim.Adam([p for p in dyn.parameters() if p.requires_grad], float(d['lr']))
    dyn = dyn.to(DEV)
    n_steps = int(d['iters']) + 1
    replay = dict(u=[[d['u%d_it%d' % (i, it)] for i in range(n - 1)] for it in range(n_steps)],
                  hard=[[d['hard%d_it%d' % (i, it)] for i in range(n - 1)] for it in range(n_steps)])
    TR.priority_tree.clear()
    np.random.seed(int(d['np_seed']))
    pm.utils.train_regressor(dyn, int(d['iters']), int(d['M']), True, opt, decoupled_reg=(mode == 'decoupled'),
                             prioritized_sampling=(mode == 'prioritized'), _replay=replay,
                             _warmup_iters=int(d['warmup']))
    for i, lin in enumerate(lins):
        assert np.allclose(lin.weight.detach().cpu().numpy(), d['W%d_final' % i], rtol=2e-3, atol=2e-5), i
        assert np.allclose(lin.bias.detach().cpu().numpy(), d['b%d_final' % i], rtol=2e-3, atol=2e-5)
    for i in range(n - 1):
        assert np.allclose(getattr(dyn.model, 'drop%d' % i).logit_p.detach().cpu().numpy(),
                           d['logit_p%d_final' % i], rtol=2e-3, atol=2e-5)
    if mode == 'prioritized':
        tree = TR.priority_tree[dyn]
        N = int(d['N'])
        assert np.array_equal(tree.counts[:N], d['tree_counts'])
        assert np.allclose(tree.sum_tree[tree.max_size - 1:tree.max_size - 1 + N], d['tree_leaves'], rtol=1e-3)


def test_iteration_is_graph_capturable():
    """A whole optimiser iteration (pack, forward sweep, rewards, loss, adjoint sweep, dW GEMM, reduce,
    device-guarded clip + Adam with its device-side step counter) records into ONE hipGraph and replays
    to the same parameters as the eager launches: nothing in it needs the host."""
    from prob_mbrl_amd import engine as E
    d = common.load('full200_nomm')
    H = int(d['H'])

    def run(n, use_graph):
        eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=DEV)
        params = args['pol_flat'].clone()
        args['pol_flat'] = params
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        loss = torch.zeros(1, device=DEV)
        step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)

        def step():
            _, _, R = eng.forward(**args)
            eng.weighted_sum(R, gw, out=loss)
            g, _, _ = eng.backward(gw)
            E.clip_adam_guarded(params, g, m, v, step_dev, 1e-3, eng.status, H, max_norm=1.0)

        if not use_graph:
            for _ in range(n):
                step()
        else:
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                step()                                   # warm-up outside the capture (1 of the n steps)
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            for _ in range(n - 1):
                graph.replay()
        torch.cuda.synchronize()
        return params.cpu().numpy().copy(), int(step_dev)

    p_eager, n_eager = run(5, False)
    p_graph, n_graph = run(5, True)
    assert n_eager == 5
    # capture itself does not execute: warm-up (1) + 4 replays = 5 steps... plus none from the capture
    assert n_graph == 5
    assert np.array_equal(p_eager, p_graph)


@pytest.mark.parametrize('name,groups', [('full200_nomm', None), ('mmg_h40', 0)])
def test_library_graph_entry_points_replay_an_iteration(name, groups, monkeypatch):
    """pmbrl_graph_capture_begin / _end / _launch: a whole optimiser iteration -- forward call with its
    queued loss, adjoint call with the device-guarded clip + Adam -- recorded through the C ABI's own entry points
    and replayed: the same parameters, bit for bit, as the eager calls.  Once on the one-launch-per-sweep form and
    once on a per-step-launch form (one moment-matching group over all rows: a launch per step), where replaying is
    what removes the launch cost."""
    from prob_mbrl_amd import engine as E
    d = dict(common.load(name))
    if groups is not None:
        d['mm_groups'] = np.asarray(groups)
        monkeypatch.setenv('PMBRL_MM_PERSTEP', '1')      # (the per-step-launch form, not the one-launch barrier form,
        monkeypatch.setenv('PMBRL_MM_PARTS', '1')        #  and not the group split over several exchanging workgroups)
    H, B = int(d['H']), d['x0'].shape[0]

    def run(n, use_graph):
        eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        gw = torch.tensor(common.loss_weights(d, B), device=DEV)
        params = args['pol_flat'].clone()
        args['pol_flat'] = params
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
        loss = eng.set_loss(gw)
        out = (torch.empty((H + 1, B, eng.D), device=DEV), torch.empty((H, B, eng.U), device=DEV),
               torch.empty((H, B, 1), device=DEV))
        adam = dict(params=params, exp_avg=m, exp_avg_sq=v, step=step_dev, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                    max_norm=1.0)

        def step():
            eng.forward(**args, out=out)
            eng.backward(gw, adam=adam)

        nodes = 0
        if not use_graph:
            for _ in range(n):
                step()
        else:
            g = E.Graph(step, warmup=1)          # one eager iteration (the warm-up), one recorded while capturing
            nodes = g.num_nodes()
            for _ in range(n - 1):
                g.replay()
        torch.cuda.synchronize()
        assert eng.valid_steps() == H
        return params.clone(), int(step_dev.item()), float(loss), nodes, eng.info

    p_e, s_e, l_e, _, info = run(4, False)
    p_g, s_g, l_g, nodes, _ = run(4, True)
    assert s_e == 4 and s_g == 4
    assert torch.equal(p_e, p_g) and l_e == l_g
    if groups == 0:
        assert info['mm_mode'] in (2, 3) and nodes >= H      # a launch per step at least: what the replay folds
    else:
        assert nodes >= 8


@pytest.mark.parametrize('name', ['full200_nomm', 'full200_mmg', 'mmg_h40'])
def test_backward_call_reports_the_loss_of_its_iteration(name, monkeypatch):
    """pmbrl_adam::loss_out_d (round 6): the fused backward call forms sum(grad_rewards * rewards) over the valid steps
    on the way of its gradient reduction and its optimiser launch writes it -- no loss launch behind the forward call.
    Against pmbrl_weighted_sum of the same rollout (another summation order: rounding only), with the same parameters
    afterwards as the form that queues the loss with the forward call (bit for bit: the loss feeds nothing); on a
    truncated horizon (status word lowered by hand: the steps beyond it do not count); and with the reduction's fusion
    switched off (PMBRL_FUSE_LOSS=0: a launch of its own inside the backward call)."""
    d = dict(common.load(name))
    H, B = int(d['H']), d['x0'].shape[0]

    def run(how, cut=None):
        eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        gw = torch.tensor(common.loss_weights(d, B), device=DEV)
        params = args['pol_flat'].clone()
        args['pol_flat'] = params
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
        loss = torch.full((1,), float('nan'), device=DEV)
        adam = dict(params=params, exp_avg=m, exp_avg_sq=v, step=step_dev, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                    max_norm=1.0, expect=min(H, 6))
        if how == 'queued':
            loss = eng.set_loss(gw)
        else:
            adam['loss_out'] = loss
        losses, refs = [], []
        for _ in range(3):
            _, _, R = eng.forward(**args)
            if cut is not None:
                eng.status[0:1].fill_(cut)
            n_valid = eng.valid_steps()
            refs.append(float((R[:n_valid, :, 0].double() * gw[:n_valid].double()).sum()))
            eng.backward(gw, adam=adam)
            losses.append(float(loss))
        torch.cuda.synchronize()
        return np.array(losses), np.array(refs), params.cpu().numpy().copy(), int(step_dev.item())

    l_q, r_q, p_q, s_q = run('queued')
    l_b, r_b, p_b, s_b = run('backward')
    assert s_q == 3 and s_b == 3
    assert np.allclose(l_b, r_b, rtol=2e-6) and np.allclose(l_q, l_b, rtol=1e-6), (l_q, l_b, r_b)
    assert np.array_equal(p_q, p_b)
    # truncated horizon: only the first 7 steps count (and the step is still taken: expect = 6)
    if H > 8:
        l_t, r_t, _, s_t = run('backward', cut=7)
        assert s_t == 3 and np.allclose(l_t, r_t, rtol=2e-6), (l_t, r_t)
        assert not np.allclose(l_t, l_b, rtol=1e-3)
    # recorded and replayed (engine.Graph): the same loss and parameters as the eager calls, bit for bit
    from prob_mbrl_amd import engine as E
    eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
    gw = torch.tensor(common.loss_weights(d, B), device=DEV)
    params = args['pol_flat'].clone()
    args['pol_flat'] = params
    m, v = torch.zeros_like(params), torch.zeros_like(params)
    step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
    loss = torch.full((1,), float('nan'), device=DEV)
    out = (torch.empty((H + 1, B, eng.D), device=DEV), torch.empty((H, B, eng.U), device=DEV),
           torch.empty((H, B, 1), device=DEV))
    adam = dict(params=params, exp_avg=m, exp_avg_sq=v, step=step_dev, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                max_norm=1.0, expect=min(H, 6), loss_out=loss)

    def step():
        eng.forward(**args, out=out)
        eng.backward(gw, adam=adam)

    g = E.Graph(step, warmup=1)
    l_g = [float(loss)]
    for _ in range(2):
        g.replay()
        l_g.append(float(loss))
    torch.cuda.synchronize()
    assert int(step_dev.item()) == 3
    assert np.array_equal(np.array(l_g), l_b) and np.array_equal(params.cpu().numpy(), p_b)
    monkeypatch.setenv('PMBRL_FUSE_LOSS', '0')
    l_u, r_u, p_u, _ = run('backward')
    assert np.allclose(l_u, r_u, rtol=2e-6) and np.array_equal(p_u, p_b)


@pytest.mark.parametrize('mode', [1, 2])
def test_repeated_calls_are_replayed_and_match_eager_calls(mode, monkeypatch):
    """pmbrl_plan_set_replay.  mode 1, the default: a per-step-launch form (one moment-matching group
    over 100 rows, a launch per step and more) replays its repeated forward and adjoint calls as hipGraphs from the third
    identical call on; mode 2: so does a one-launch form when asked.  The parameters after 6 optimiser iterations are
    those of the eager calls bit for bit; a call with other arguments in between (another output buffer) runs eagerly
    and the replay resumes after it."""
    name = 'mmg_h40' if mode == 1 else 'full200_nomm'
    d = dict(common.load(name))
    if mode == 1:
        d['mm_groups'] = np.asarray(0)
        monkeypatch.setenv('PMBRL_MM_PERSTEP', '1')
        monkeypatch.setenv('PMBRL_MM_PARTS', '1')
    H, B = int(d['H']), d['x0'].shape[0]

    def run(replay):
        eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        eng.set_replay(replay)
        assert eng.info['replay'] == (1 if replay else 0)
        gw = torch.tensor(common.loss_weights(d, B), device=DEV)
        params = args['pol_flat'].clone()
        args['pol_flat'] = params
        m, v = torch.zeros_like(params), torch.zeros_like(params)
        step_dev = torch.zeros(1, dtype=torch.int64, device=DEV)
        loss = eng.set_loss(gw)
        adam = dict(params=params, exp_avg=m, exp_avg_sq=v, step=step_dev, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, max_norm=1.0)
        other = (torch.empty((H + 1, B, eng.D), device=DEV), torch.empty((H, B, eng.U), device=DEV),
                 torch.empty((H, B, 1), device=DEV))
        losses = []
        for it in range(7):
            if it == 4:       # other output tensors: not the recorded call
                eng.forward(**args, out=other)
            else:
                eng.forward(**args)
            eng.backward(gw, adam=adam)
            losses.append(float(loss))
        torch.cuda.synchronize()
        assert eng.valid_steps() == H
        return params.clone(), int(step_dev.item()), losses, eng.replay_count(), eng.info

    p_e, s_e, l_e, n_e, info = run(0)
    p_r, s_r, l_r, n_r, _ = run(mode)
    if mode == 1:
        assert info['mm_mode'] in (2, 3)
    assert n_e == (0, 0)
    # forward: calls 0, 1 eager (1 = the recording, launched as a graph), 2, 3 replayed, 4 eager (other outputs), 5 eager
    # (first sight again), 6 recorded + launched; the adjoint's arguments change with the forward's outputs at 4 and 5
    assert n_r[0] >= 3 and n_r[1] >= 3, n_r
    assert s_e == 7 and s_r == 7
    assert l_e == l_r
    assert torch.equal(p_e, p_r)
