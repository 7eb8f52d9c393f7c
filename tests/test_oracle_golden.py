"""CPU: the oracle (oracle/ref_torch.py, oracle/adjoint_np.py) against the golden
vectors generated from the real reference (tools/make_golden.py)."""
import numpy as np
import pytest
import torch

from oracle import adjoint_np as A
from oracle import ref_torch as R
from tests import common


@pytest.mark.parametrize('name', common.fixture_names('iter'))
def test_torch_oracle_matches_reference_fp32(name):
    d = common.load(name)
    torch.set_flush_denormal(True)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float32)
    loss, g, (S, Ac, Rw) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, meta['maximize'],
                                       meta['mm_states'], meta['mm_rewards'], meta['mm_groups'],
                                       z_mm, z_rr, meta['infer_ns'])
    # same op sequence as the reference -> fp32 agreement to rounding of the BLAS calls
    assert np.allclose(torch.stack(S).detach().numpy(), d['ref32_states'], rtol=1e-5, atol=1e-6)
    assert np.allclose(torch.stack(Ac).detach().numpy(), d['ref32_actions'], rtol=1e-5, atol=1e-6)
    assert np.allclose(torch.stack(Rw).detach().numpy().reshape(d['ref32_rewards'].shape),
                       d['ref32_rewards'], rtol=1e-5, atol=1e-7)
    assert abs(float(loss) - float(d['ref32_loss'])) <= 1e-5 * abs(float(d['ref32_loss']))
    if 'ref32_grad' in d:      # (wide fixtures keep the fp64 gradient only)
        # ill-conditioned moment matching: two fp32 evaluations of the same formulas differ by about as
        # much as the reference's fp32 run differs from its fp64 run
        assert common.rel(g.numpy(), d['ref32_grad']) < 1e-4 + 2 * common.rel(d['ref32_grad'], d['ref64_grad'])


@pytest.mark.parametrize('name', common.fixture_names('iter'))
def test_torch_oracle_matches_reference_fp64(name):
    d = common.load(name)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    loss, g, (S, Ac, Rw) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, meta['maximize'],
                                       meta['mm_states'], meta['mm_rewards'], meta['mm_groups'],
                                       z_mm, z_rr, meta['infer_ns'])
    assert common.rel(torch.stack(S).detach().numpy(), d['ref64_states']) < 1e-6
    assert abs(float(loss) - float(d['ref64_loss'])) <= 1e-7 * abs(float(d['ref64_loss']))
    assert common.rel(g.numpy(), d['ref64_grad']) < 1e-6


@pytest.mark.parametrize('name', [n for n in common.fixture_names('iter') if not n.startswith('gmm_')])
def test_explicit_adjoint_matches_reference_fp64(name):
    """The autograd-free forward/adjoint the HIP kernels implement."""
    d = common.load(name)
    P = A.Problem(d, np.float64)
    st = A.forward(P)
    g, gx0, _ = A.backward(P, st)
    assert common.rel(np.stack(st['states']), d['ref64_states']) < 1e-6
    assert common.rel(np.stack(st['rewards']).reshape(d['ref64_rewards'].shape),
                      d['ref64_rewards']) < 1e-6
    assert abs(A.loss(P, st) - float(d['ref64_loss'])) <= 1e-7 * abs(float(d['ref64_loss']))
    assert common.rel(g, d['ref64_grad']) < 1e-6


@pytest.mark.parametrize('name', common.fixture_names('mcp'))
def test_oracle_mc_pilco_iterations(name):
    """Multi-iteration fixture from the REAL algorithms.mc_pilco: pins clip + Adam."""
    d = common.load(name)
    torch.set_flush_denormal(True)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float32)
    params = R.policy_params(pol)
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    losses = []
    for it in range(int(d['mcp_n_iters'])):
        loss, g, _ = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                 meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr,
                                 cvar_eps=float(d['mcp_cvar_eps']) if 'mcp_cvar_eps' in d else 0.0,
                                 reg_weight=float(d['mcp_reg_weight']) if 'mcp_reg_weight' in d else 0.0)
        losses.append(float(loss))
        grads = [p.grad for p in params]
        _, grads = R.clip_grad_norm(grads, float(d['mcp_clip']))
        with torch.no_grad():
            for p, gg, m, v in zip(params, grads, ms, vs):
                R.adam_step(p, gg, m, v, it + 1, float(d['mcp_lr']))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=2e-5)
    final = torch.cat([p.detach().reshape(-1) for p in params]).numpy()
    assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=1e-6)


def test_oracle_truncated_horizon():
    """utils/rollout.py:154-157: after a failure in step n > 5 the reference optimises on the first n
    steps; fixture from the reference's own rollout with a RuntimeError raised in step 8 of 12."""
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    for dt, tag, tol in ((torch.float32, 'ref32_', 1e-4), (torch.float64, 'ref64_', 1e-6)):
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, dt)
        loss, g, (S, Ac, Rw) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True,
                                           meta['mm_groups'], z_mm, z_rr, n_steps=n)
        assert len(Rw) == n and d[tag + 'rewards'].shape[0] == n and d[tag + 'states'].shape[0] == n + 1
        assert common.rel(torch.stack(S[:n]).detach().numpy(), d[tag + 'states'][:n]) < tol
        assert abs(float(loss) - float(d[tag + 'loss'])) <= tol * abs(float(d[tag + 'loss']))
        assert common.rel(g.numpy(), d[tag + 'grad']) < tol
    P = A.Problem(d, np.float64)
    P.H = n
    st = A.forward(P)
    g, _, _ = A.backward(P, st)
    assert common.rel(g, d['ref64_grad']) < 1e-6


def _adam_loop(d, x0_of, extra_of, after=None):
    """clip + Adam iterations of the oracle; x0_of(it) / extra_of(it) give the start states and
    the extra iteration() arguments, after(it, traj) sees the trajectory."""
    torch.set_flush_denormal(True)
    _, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float32)
    params = R.policy_params(pol)
    ms = [torch.zeros_like(p) for p in params]
    vs = [torch.zeros_like(p) for p in params]
    losses = []
    for it in range(int(d['mcp_n_iters'])):
        loss, g, traj = R.iteration(x0_of(it), pol, dyn, spec, meta['H'], gamma, True,
                                    meta['mm_states'], meta['mm_rewards'], meta['mm_groups'], z_mm,
                                    z_rr, **extra_of(it))
        losses.append(float(loss))
        _, grads = R.clip_grad_norm([p.grad for p in params], float(d['mcp_clip']))
        with torch.no_grad():
            for p, gg, m, v in zip(params, grads, ms, vs):
                R.adam_step(p, gg, m, v, it + 1, float(d['mcp_lr']))
        if after is not None:
            after(it, traj)
    return losses, torch.cat([p.detach().reshape(-1) for p in params]).numpy()


def test_oracle_value_bootstrap():
    """mc_pilco(value_func=V) of the reference: the terminal value enters the return with
    discount(H) (algorithms/mc_pilco.py:136-140)."""
    d = common.load('ext_value')
    x0 = torch.tensor(d['x0'])
    val = R.value_from_npz(d)
    H = int(d['H'])
    losses, final = _adam_loop(d, lambda it: x0, lambda it: dict(value=val, gamma_H=1.0 / H))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=2e-5)
    assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=1e-6)
    # the bootstrap matters: without it the loss is a different number
    l0, _ = _adam_loop(d, lambda it: x0, lambda it: {})
    assert abs(l0[0] - float(d['ref32_mcp_losses'][0])) > 1e-3 * abs(l0[0])


def test_oracle_prioritized_replay():
    """mc_pilco(prioritized_replay=True) of the reference (algorithms/mc_pilco.py:80-84, 156-188,
    222-246): the host-side SumTree + the oracle's iteration reproduce the sampled start states of
    every iteration, the losses, the final parameters and the final priorities."""
    from prob_mbrl_amd.experience import SumTree
    d = common.load('ext_replay')
    tree = SumTree(2**20)
    B = d['x0'].shape[0]
    state = dict(x0=torch.tensor(d['x0']), idxs=None, w=None, beta=0.4)
    np.random.seed(int(d['replay_np_seed']))
    x0s = []

    def x0_of(it):
        x0s.append(state['x0'].numpy().copy())
        return state['x0']

    def extra_of(it):
        if state['idxs'] is None:
            return dict(want_action_norms=True)
        return dict(is_weights=state['w'], want_action_norms=True)

    def after(it, traj):
        if state['idxs'] is not None:
            scores = traj[3].mean(0).numpy() / tree.counts[state['idxs'] - tree.max_size + 1]
            for idx, p in zip(state['idxs'], (scores + 1e-8)**0.6):
                tree.update(idx, p)
            tree.renormalize()
        if it == 0:
            for e in range(int(d['replay_n_episodes'])):
                for x in d['replay_states%d' % e]:
                    tree.append(x, tree.max_p)
                    tree.renormalize()
        xs, state['idxs'], w = tree.sample(B, beta=state['beta'])
        state['beta'] = max(1.0, state['beta'] + 0.1)
        state['x0'] = torch.tensor(np.stack(xs), dtype=torch.float32)
        state['w'] = torch.tensor(np.stack(w), dtype=torch.float32)

    losses, final = _adam_loop(d, x0_of, extra_of, after)
    assert np.array_equal(np.stack(x0s), d['replay_x0s'].astype(np.float32))
    assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
    assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=1e-6)
    n = len(d['replay_final_counts'])
    assert np.array_equal(tree.counts[:n], d['replay_final_counts'])
    assert np.allclose(tree.sum_tree[tree.max_size - 1:tree.max_size - 1 + n], d['replay_final_leaves'],
                       rtol=1e-4)


def test_oracle_critic_fit():
    """update_value_function of the reference's example script (critic without an output density),
    three Adam updates replayed with the recorded Bernoulli outcomes."""
    d = common.load('critic_fit')
    torch.set_flush_denormal(True)
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32))  # noqa: E731
    n = int(d['n_layers'])
    weights = [(T(d['W%d_init' % i]).requires_grad_(True), T(d['b%d_init' % i]).requires_grad_(True)) for i in range(n)]
    logit_ps = [T(d['logit_p%d_init' % i]).requires_grad_(True) for i in range(n - 1)]
    params = []
    for i in range(n):
        params += list(weights[i]) + ([logit_ps[i]] if i < n - 1 else [])
    ms, vs = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    norm = {k: T(d[k]).reshape(1, -1) for k in ('mx', 'iSx', 'my', 'Sy')}
    rewards = [T(r) for r in d['rewards']]
    for it in range(int(d['n_updates'])):
        for p in params:
            p.grad = None
        hards = [T(d['hard%d_it%d' % (k, it)]) for k in range(2 * (n - 1))]
        loss = R.critic_update(weights, logit_ps, [float(d['temp%d' % i]) for i in range(n - 1)],
                               [float(d['reg_scale%d' % i]) for i in range(n - 1)],
                               [float(d['drop_reg%d' % i]) for i in range(n - 1)],
                               [T(d['u%d' % i]) for i in range(n - 1)], norm, T(d['states0']), T(d['statesH']),
                               rewards, float(d['gamma']), int(d['H']), hards, float(d['reg_weight']))
        loss.backward()
        with torch.no_grad():
            for p, m, v in zip(params, ms, vs):
                R.adam_step(p, p.grad, m, v, it + 1, float(d['lr']))
        for i in range(n):
            assert np.allclose(weights[i][0].detach().numpy(), d['W%d_it%d' % (i, it)], rtol=2e-4, atol=2e-6)
            assert np.allclose(weights[i][1].detach().numpy(), d['b%d_it%d' % (i, it)], rtol=2e-4, atol=2e-6)
        for i in range(n - 1):
            assert np.allclose(logit_ps[i].detach().numpy(), d['logit_p%d_it%d' % (i, it)], rtol=2e-4, atol=2e-6)


def test_tile_layout():
    x = torch.arange(6.).view(3, 2)
    t = R.tile(x, 4)
    assert t.shape == (12, 2)
    for g in range(3):
        for k in range(4):
            assert torch.equal(t[g * 4 + k], x[g])


@pytest.mark.parametrize('name', common.fixture_names('standalone'))
def test_oracle_standalone_forwards_match_reference(name):
    """Policy.forward / DynamicsModel.forward outside a rollout (models/core.py:221-248, 265-303)."""
    d = common.load(name)
    for dt, tag, tol in ((torch.float32, 'ref32_', 2e-6), (torch.float64, 'ref64_', 3e-7)):   # the reference's .double() run keeps a few fp32-rounded constants
        x, pol, dyn, spec, _, _, _, _ = R.problem_from_npz(d, dt)
        a = R.policy_forward(x, pol)
        assert np.allclose(a.numpy(), d[tag + 'act'], rtol=tol, atol=tol)
        nxt, rew = R.dynamics_forward(x, a, dyn, spec)
        assert np.allclose(nxt.numpy(), d[tag + 'next'], rtol=tol, atol=tol)
        assert np.allclose(rew.numpy().reshape(-1, 1), d[tag + 'rew'], rtol=tol, atol=tol)
        assert np.allclose((nxt - x).numpy(), d[tag + 'delta'], rtol=10 * tol, atol=10 * tol)


@pytest.mark.parametrize('name', common.fixture_names('bnn'))
def test_oracle_bnn_training_matches_reference(name):
    """utils.train_regressor's iteration body (forward in train() mode with concrete dropout,
    Gaussian NLL, dropout regulariser, Adam) replayed with the recorded random draws."""
    import numpy as np
    d = np.load(common.os.path.join(common.GOLDEN, name + '.npz'))
    params, losses, g0 = R.bnn_train(d, torch.float32)
    nl = int(d['n_layers'])
    assert np.allclose(np.array(losses), d['losses'], rtol=2e-5, atol=1e-6)
    k = 0
    for l in range(nl):
        for key in ('W%d' % l, 'b%d' % l) + (('logit_p%d' % l,) if l < nl - 1 else ()):
            gkey = 'g' + key + '_it0'
            assert common.rel(g0[k].numpy(), d[gkey]) < 2e-5, gkey
            assert np.allclose(params[k].detach().numpy(), d[key + '_final'], rtol=1e-5, atol=2e-7), key
            k += 1
