"""GPU (-m gpu): the HIP path through the C ABI against the oracle / golden
fixtures.  Tolerances: the north star asks for 1e-4 relative on trajectory cost
and policy gradient; the tests hold the kernels to 2e-5 against the fp64
reference values (fp32 MFMA + fp32 transcendentals)."""
import os

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu

TOL_TRAJ = 2e-5
TOL_GRAD = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available(), 'needs a HIP device'
    return torch.device('cuda:0')


@pytest.mark.parametrize('R,K,O', [(16, 4, 200), (16, 200, 200), (16, 200, 2), (7, 5, 33),
                                   (32, 200, 8), (32, 208, 200), (32, 512, 512), (64, 40, 64),
                                   (16, 16, 16), (48, 130, 70)])
@pytest.mark.parametrize('transpose', [False, True])
def test_mfma_linear(dev, R, K, O, transpose):
    """y = x W^T + b through gemm_tiles / gemm_narrow; asymmetric random operands
    catch row/col swaps of the MFMA fragment layout."""
    from prob_mbrl_amd import engine as E
    g = torch.Generator().manual_seed(R * 1000 + K * 10 + O)
    x = torch.randn(R, K, generator=g)
    W = torch.randn(O, K, generator=g)
    b = torch.randn(O, generator=g)
    ref = (x.double() @ W.double().t() + b.double()).numpy()
    Wd = W.t().contiguous() if transpose else W
    y = E.debug_linear(x.to(dev), Wd.to(dev), b.to(dev), transpose_w=transpose).cpu().numpy()
    scale = np.abs(x.numpy()) @ np.abs(W.numpy()).T + np.abs(b.numpy())
    assert np.all(np.abs(y - ref) <= 2e-6 * scale + 1e-6), np.abs(y - ref).max()


def test_pack_mask(dev):
    from prob_mbrl_amd import engine as E
    g = torch.Generator().manual_seed(3)
    for B, h in [(5, 7), (40, 200), (33, 16), (2500, 200)]:
        m = (torch.rand(B, h, generator=g) < 0.9).float()
        bits = E.pack_mask(m.to(dev)).cpu().numpy().view(np.uint16)
        nt = (h + 15) // 16
        pad = np.zeros((B, nt * 16), dtype=np.uint8)
        pad[:, :h] = m.numpy().astype(np.uint8)
        want = np.packbits(pad, axis=1, bitorder='little').view(np.uint16)
        assert np.array_equal(bits, want)


def test_weighted_sum(dev):
    from prob_mbrl_amd import engine as E
    a = torch.randn(40, 2500, device=dev)
    w = torch.randn(40, 2500, device=dev)
    eng_sum = E.Engine.weighted_sum
    out = torch.empty(1, device=dev)
    from prob_mbrl_amd import _lib
    _lib.check(_lib.load().pmbrl_weighted_sum(E._stream(), E._ptr(a), E._ptr(w), a.numel(),
                                              E._ptr(out)), 'ws')
    want = float((a.double() * w.double()).sum())
    assert abs(float(out) - want) <= 1e-6 * float((a.double() * w.double()).abs().sum())


@pytest.mark.parametrize('n', [1, 777, 41602, 550416])
@pytest.mark.parametrize('clip', [None, 1.0, 1e-3])
def test_clip_adam_matches_torch(dev, n, clip):
    from prob_mbrl_amd import engine as E
    torch.manual_seed(n)
    p0 = torch.randn(n, device=dev)
    p_ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([p_ref], lr=1e-3)
    p = p0.clone()
    m = torch.zeros_like(p)
    v = torch.zeros_like(p)
    for step in range(1, 4):
        g = torch.randn(n, device=dev) * (10.0 if step == 2 else 0.1)
        p_ref.grad = g.clone()
        if clip is not None:
            torch.nn.utils.clip_grad_norm_([p_ref], clip)
        opt.step()
        gg = g.clone()
        norm = torch.empty(1, device=dev)
        E.clip_adam(p, gg, m, v, step, 1e-3, max_norm=clip, norm_out=norm)
        assert abs(float(norm) - float(g.double().norm())) <= 1e-5 * float(g.double().norm())
        assert torch.allclose(gg, p_ref.grad, rtol=1e-5, atol=1e-8)
        assert torch.allclose(p, p_ref.detach(), rtol=1e-5, atol=1e-6), (p - p_ref).abs().max()
    st = opt.state[p_ref]
    assert torch.allclose(m, st['exp_avg'], rtol=1e-5, atol=1e-7)
    assert torch.allclose(v, st['exp_avg_sq'], rtol=1e-5, atol=1e-12)


def _wide(d):
    """hidden layers wider than the latency-optimised family takes (> 16 tiles): general kernels only"""
    return max(d['pol_W0'].shape[0], d['dyn_W0'].shape[0]) > 256


def _run(d, dev, hint=0, generic=False, n_valid=None, precision=None):
    eng, args, _ = common.engine_from_fixture(d, dev, rows_per_wg_hint=hint, force_generic=generic,
                                              precision=precision)
    assert bool(eng.info['fast']) == (not generic and not _wide(d) and not bool(d.get('infer_ns', False))
                                      and np.asarray(d['pol_mask0']).ndim == 2)
    if precision in ('split', 'split_f16'):
        assert eng.info['precision'] == precision      # the bf16 / fp16 matrix-core path really ran
    S, A, Rw = eng.forward(**args)
    if n_valid is not None:     # pretend the sweep failed at step n_valid (see test_truncated_horizon)
        eng.status[0] = n_valid
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    loss = eng.weighted_sum(Rw, gw)
    g, gx0, agn = eng.backward(gw, want_x0=True, want_agn=True)
    torch.cuda.synchronize()
    return eng, S.cpu().numpy(), A.cpu().numpy(), Rw.cpu().numpy(), float(loss), \
        g.cpu().numpy().copy(), gx0.cpu().numpy(), agn.cpu().numpy()


# (the C5 shape, infer_noise_variables, per-step dropout masks and angle_dims inside the networks exist in the
# general family only)
_PARITY_CASES = [(n, g) for n in common.fixture_names('iter') for g in (False, True)
                 if g or not (n.startswith(('c5_', 'stepmask', 'angles_', 'gmm_')) or 'infer_ns' in n)]


@pytest.mark.parametrize('name,generic', _PARITY_CASES,
                         ids=['%s-%s' % (n, 'generic' if g else 'fast') for n, g in _PARITY_CASES])
def test_rollout_parity(dev, name, generic):
    """Both kernel families (latency-optimised 'fast' and the generic one) vs the fixtures."""
    d = common.load(name)
    eng, S, A, Rw, loss, g, gx0, agn = _run(d, dev, generic=generic)
    assert eng.valid_steps() == int(d['H'])
    assert common.rel(S, d['ref64_states']) < TOL_TRAJ
    assert common.rel(A, d['ref64_actions']) < TOL_TRAJ
    assert common.rel(Rw.reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    # trajectory cost: vs the fp64 reference and vs the reference's own fp32 run
    assert abs(loss - float(d['ref64_loss'])) <= TOL_TRAJ * abs(float(d['ref64_loss']))
    assert abs(loss - float(d['ref32_loss'])) <= 1e-4 * abs(float(d['ref32_loss']))
    # policy gradient
    assert common.rel(g, d['ref64_grad']) < TOL_GRAD
    if 'ref32_grad' in d:
        assert common.rel(g, d['ref32_grad']) < TOL_GRAD + common.rel(d['ref32_grad'], d['ref64_grad'])


def _reg_fixtures():
    """fixtures whose shape the register-resident family takes (three-layer nets of hidden width 177..208, D + U <= 8)"""
    out = []
    for n in common.fixture_names('iter'):
        d = np.load(os.path.join(common.GOLDEN, n + '.npz'))
        if ('pol_W1' in d and int(d['pol_n_layers']) == 3 and int(d['dyn_n_layers']) == 3 and
                176 < d['pol_W0'].shape[0] <= 208 and d['pol_W0'].shape[0] == d['dyn_W0'].shape[0]):
            out.append(n)
    return out


@pytest.mark.parametrize('name', _reg_fixtures())
def test_rollout_parity_lean(dev, name):
    """The HEADLINE kernels against reference-generated vectors: the plain call -- no optional outputs -- is what
    bench.py and mc_pilco's fused iteration issue, and what the register-resident family (csrc/pmbrl_reg.h:
    pm_reg_fwd_kernel / pm_reg_bwd_kernel) serves; test_rollout_parity above asks for dL/dx0 and the action-gradient
    norms, which routes its adjoint to the latency-optimised family."""
    d = common.load(name)
    eng, args, _ = common.engine_from_fixture(d, dev)
    if not eng.info['reg']:
        pytest.skip('the register-resident family does not take this configuration (moment matching: %s)' % bool(d['mm_states']))
    S, A, Rw = eng.forward(**args)
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    loss = float(eng.weighted_sum(Rw, gw))
    g, _, _ = eng.backward(gw)
    torch.cuda.synchronize()
    assert eng.reg_calls() == (1, 1), eng.reg_calls()          # both sweeps ran on pm_reg_*
    assert eng.valid_steps() == int(d['H'])
    assert common.rel(S.cpu().numpy(), d['ref64_states']) < TOL_TRAJ
    assert common.rel(A.cpu().numpy(), d['ref64_actions']) < TOL_TRAJ
    assert common.rel(Rw.cpu().numpy().reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    assert abs(loss - float(d['ref64_loss'])) <= TOL_TRAJ * abs(float(d['ref64_loss']))
    assert common.rel(g.cpu().numpy(), d['ref64_grad']) < TOL_GRAD
    # ... and the other family from the same stashes: same trajectory bit for bit is not promised (different K order),
    # the gradient agrees far inside the bar
    g_reg = g.cpu().numpy().copy()
    g2, _, _ = eng.backward(gw, want_x0=True, want_agn=True)
    torch.cuda.synchronize()
    assert eng.reg_calls() == (1, 1)                           # that call did NOT run on the family
    assert common.rel(g2.cpu().numpy(), g_reg) < 2e-5


@pytest.mark.parametrize('name', [n for n, g in _PARITY_CASES if g])
def test_rollout_parity_general_family_fp32(dev, name):
    """The general kernel family in exact fp32 (PMBRL_PREC_F32): test_rollout_parity above runs it in the default
    arithmetic (split operands), this keeps the fp32 instantiations under the same fixtures and tolerances."""
    d = common.load(name)
    eng, S, A, Rw, loss, g, gx0, agn = _run(d, dev, generic=True, precision='f32')
    assert eng.info['precision'] == 'f32' and eng.valid_steps() == int(d['H'])
    assert common.rel(S, d['ref64_states']) < TOL_TRAJ
    assert common.rel(A, d['ref64_actions']) < TOL_TRAJ
    assert common.rel(Rw.reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    assert abs(loss - float(d['ref64_loss'])) <= TOL_TRAJ * abs(float(d['ref64_loss']))
    assert common.rel(g, d['ref64_grad']) < TOL_GRAD


def test_mixture_head_exact_noise_gradient(dev):
    """PMBRL_FLAG_GMM_EXACT_NOISE_GRAD: the noise term of the mixture head differentiated with each step's own
    noise (the mathematically exact gradient) instead of the reference's last-step noise; against the torch oracle
    in that mode."""
    from oracle import ref_torch as R
    d = common.load('gmm_d4')
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    dyn['exact_noise_grad'] = True
    loss64, g64, _ = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, meta['maximize'], meta['mm_states'],
                                 meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr, meta['infer_ns'])
    assert common.rel(g64.numpy(), d['ref64_grad']) > 1e-2           # (it IS a different gradient)
    from prob_mbrl_amd import problem as PB
    import prob_mbrl_amd.engine as E
    orig = E.Engine.__init__

    def init(self, *a, **k):
        k['gmm_exact_noise_grad'] = True
        orig(self, *a, **k)
    E.Engine.__init__ = init
    try:
        eng, args, _ = PB.engine_from_problem(d, dev)
    finally:
        E.Engine.__init__ = orig
    S, A, Rw = eng.forward(**args)
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    g, _, _ = eng.backward(gw)
    assert common.rel(S.cpu().numpy(), d['ref64_states']) < TOL_TRAJ
    assert common.rel(g.cpu().numpy(), g64.numpy()) < TOL_GRAD


_SPLIT_CASES = [n for n, g in _PARITY_CASES if not g]


@pytest.mark.parametrize('prec', ['split', 'split_f16'])
@pytest.mark.parametrize('name', _SPLIT_CASES)
def test_rollout_parity_split_precision(dev, name, prec):
    """PMBRL_PREC_SPLIT / _SPLIT_F16: the hidden-width GEMMs on the bf16 / fp16 matrix cores with split
    operands (forward: three bf16 or two fp16 pieces = fp32-equivalent; adjoint: two bf16 pieces) against
    the same fixtures and the same tolerances as the exact-fp32 path.  Reports the errors next to the
    fp32 path's."""
    d = common.load(name)
    eng, S, A, Rw, loss, g, gx0, agn = _run(d, dev, precision=prec)
    _, S0, A0, Rw0, loss0, g0, _, _ = _run(d, dev, precision='f32')
    assert eng.valid_steps() == int(d['H'])
    e_s, e_g = common.rel(S, d['ref64_states']), common.rel(g, d['ref64_grad'])
    print('%s %s: states %.2e grad %.2e | f32: states %.2e grad %.2e' %
          (name, prec, e_s, e_g, common.rel(S0, d['ref64_states']), common.rel(g0, d['ref64_grad'])))
    assert e_s < TOL_TRAJ
    assert common.rel(A, d['ref64_actions']) < TOL_TRAJ
    assert common.rel(Rw.reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    assert abs(loss - float(d['ref64_loss'])) <= TOL_TRAJ * abs(float(d['ref64_loss']))
    assert e_g < TOL_GRAD


@pytest.mark.parametrize('generic', [False, True], ids=['fast', 'generic'])
def test_truncated_horizon(dev, generic):
    """utils/rollout.py:154-157: after a failure at step n > 5 the reference optimises on the first n
    steps.  Fixture from the reference's own rollout (RuntimeError raised in step 8 of 12).  Here the
    forward sweep completes; the status word is then set to 8, which is all the loss kernel, the
    adjoint sweep and the dW GEMM look at -- and the stashes of the steps >= 8 are poisoned with NaN
    to prove nothing reads them (they hold non-finite values after a real failure)."""
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    eng, args, _ = common.engine_from_fixture(d, dev, force_generic=generic)
    S, A, Rw = eng.forward(**args)
    eng.status[0] = n
    S[n + 1:] = float('nan')
    A[n:] = float('nan')
    Rw[n:] = float('nan')
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    gw_p = gw.clone()
    gw_p[n:] = float('nan')
    loss = float(eng.weighted_sum(Rw, gw_p))
    g, gx0, agn = eng.backward(gw_p, want_x0=True, want_agn=True)
    g = g.cpu().numpy().copy()
    assert common.rel(S[:n + 1].cpu().numpy(), d['ref64_states']) < TOL_TRAJ
    assert common.rel(Rw[:n].cpu().numpy().reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    assert abs(loss - float(d['ref64_loss'])) <= TOL_TRAJ * abs(float(d['ref64_loss']))
    assert np.all(np.isfinite(g)) and common.rel(g, d['ref64_grad']) < TOL_GRAD
    assert torch.isfinite(gx0).all() and torch.isfinite(agn[:n]).all()
    # the full horizon of the same problem is a different gradient
    eng.forward(**args)
    g_full = eng.backward(gw)[0].cpu().numpy()
    assert common.rel(g_full, d['ref64_grad']) > 1e-2


@pytest.mark.parametrize('generic', [False, True], ids=['fast', 'generic'])
@pytest.mark.parametrize('name', ['nomm_d4', 'mmg_d4', 'full200_nomm', 'rdv_d8_u4_3layer'])
def test_grad_x0_and_action_norms(dev, name, generic):
    """dL/dx0 and the per-step ||dL/da_t|| (prioritised replay hook) vs the explicit adjoint."""
    from oracle import adjoint_np as ADJ
    d = common.load(name)
    _, S, A, Rw, loss, g, gx0, agn = _run(d, dev, generic=generic)
    P = ADJ.Problem(d, np.float64)
    st = ADJ.forward(P)
    g_ref, gx0_ref, Gst = ADJ.backward(P, st)
    assert common.rel(g, g_ref) < TOL_GRAD
    assert common.rel(gx0, gx0_ref) < TOL_GRAD


@pytest.mark.parametrize('name,hint', [('nomm_d4', 32), ('nomm_d4', 64), ('full200_nomm', 32),
                                       ('full200_nomm', 64), ('mmg_d4', 16), ('mmg_d4', 32),
                                       ('rdv_d8_u4_3layer', 64)])
@pytest.mark.parametrize('generic', [False, True], ids=['fast', 'generic'])
def test_row_tile_variants(dev, name, hint, generic):
    d = common.load(name)
    eng, S, A, Rw, loss, g, gx0, agn = _run(d, dev, hint, generic)
    assert common.rel(S, d['ref64_states']) < TOL_TRAJ
    assert common.rel(g, d['ref64_grad']) < TOL_GRAD


@pytest.mark.parametrize('name', ['nomm_d4', 'mmg_d4', 'full200_mmg'])
def test_deterministic(dev, name):
    d = common.load(name)
    r1 = _run(d, dev)
    r2 = _run(d, dev)
    assert np.array_equal(r1[1], r2[1]) and np.array_equal(r1[5], r2[5])


@pytest.mark.parametrize('name,world', [('nomm_d4', 2), ('nomm_d4', 4), ('mmg_d4', 2),
                                        ('mmg_d4', 4), ('full200_mmg', 2)])
def test_sharded_rows_reproduce_full_gradient(dev, name, world):
    """Multi-GPU decomposition on one device: rank r owns a contiguous block of
    rows (whole groups), gradients are summed on the host ("fake collective")."""
    d = common.load(name)
    B = d['x0'].shape[0]
    gw_full = common.loss_weights(d, B)
    total = None
    states = []
    for rank in range(world):
        eng, args, (lo, hi) = common.engine_from_fixture(d, dev, shard=(rank, world))
        S, A, Rw = eng.forward(**args)
        g, _, _ = eng.backward(torch.tensor(gw_full[:, lo:hi].copy(), device=dev))
        g = g.cpu().numpy().astype(np.float64)
        total = g if total is None else total + g
        states.append(S.cpu().numpy())
    S = np.concatenate(states, axis=1)
    assert common.rel(S, d['ref64_states']) < TOL_TRAJ
    assert common.rel(total, d['ref64_grad']) < TOL_GRAD


def test_failure_is_reported_not_nan(dev):
    """A rank-deficient moment-matching group (M <= D rows) must surface as a step
    index in the status word (-> RuntimeError upstream), never as silent NaNs."""
    d = dict(common.load('mmg_d4'))
    d['mm_groups'] = np.asarray(20)          # 40 rows -> M = 2 rows per group, D = 4
    eng, args, _ = common.engine_from_fixture(d, dev)
    eng.forward(**args)
    assert eng.valid_steps() == 0


@pytest.mark.gpu
@pytest.mark.parametrize('B,dims', [(1, [5, 200, 200, 2]), (37, [9, 48, 24, 40, 8]), (100, [4, 16, 2])])
def test_mlp_forward_matches_torch(B, dims):
    """pmbrl_mlp_forward (stand-alone BNN + Gaussian head) against a plain torch fp32 evaluation
    of the same formula (models/core.py:169-187, models/densities.py:87-121)."""
    from prob_mbrl_amd import engine as E
    g = torch.Generator().manual_seed(B + len(dims))
    dev = torch.device('cuda:0')
    nl = len(dims) - 1
    n_out = dims[-1] // 2
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i]) for i in range(nl)]
    bs = [0.1 * torch.randn(dims[i + 1], generator=g) for i in range(nl)]
    masks = [(torch.rand(B, dims[i + 1], generator=g) < 0.8).float() for i in range(nl - 1)]
    masks[0] = None if nl > 2 else masks[0]          # one layer without dropout
    keep = [0.8 if m is not None else 1.0 for m in masks]
    x = torch.randn(B, dims[0], generator=g)
    z = torch.randn(B, n_out, generator=g)
    shift, iscale = torch.randn(dims[0], generator=g), 0.5 + torch.rand(dims[0], generator=g)
    osc, osh = 0.5 + torch.rand(n_out, generator=g), torch.randn(n_out, generator=g)
    sqs, sqb = 1.0 + torch.rand(n_out, generator=g), 0.1 * torch.randn(n_out, generator=g)
    h = (x - shift) * iscale
    for i in range(nl - 1):
        h = torch.relu(h @ Ws[i].t() + bs[i])
        if masks[i] is not None:
            h = h * masks[i] / keep[i]
    o = h @ Ws[-1].t() + bs[-1]
    mu, ls = o[:, :n_out], o[:, n_out:]
    mls = float(np.log(5.0))
    ls = -torch.nn.functional.softplus(-ls + mls) + mls + osc.log()
    mu = mu * osc + osh
    want = sqs * torch.tanh(mu + z * ls.exp()) + sqb
    flat = torch.cat([t.reshape(-1) for pair in zip(Ws, bs) for t in pair]).to(dev)
    bits = [E.pack_mask(m.to(dev)) if m is not None else None for m in masks]
    out = E.mlp_forward(x.to(dev), flat, dims, keep, bits, z.to(dev), shift, iscale, osc, osh, sqs, sqb,
                        max_log_std=mls, want=('sample', 'mean', 'log_std'))
    assert np.allclose(out['mean'].cpu().numpy(), mu.numpy(), rtol=2e-5, atol=2e-6)
    assert np.allclose(out['log_std'].cpu().numpy(), ls.numpy(), rtol=2e-5, atol=2e-6)
    assert np.allclose(out['sample'].cpu().numpy(), want.numpy(), rtol=2e-5, atol=2e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('name', [n for n in common.fixture_names('bnn') if n != 'bnn_gmm'])
def test_bnn_fused_training_matches_reference(name):
    """pmbrl_bnn_train_steps -- whole iterations in two launches each (forward + backward; dW, regulariser, Adam, re-packed
    weights and loss in one kernel), ALL of the fixture's iterations queued by one call -- replaying the reference's
    train_regressor run (recorded minibatches and dropout draws): every iteration's loss and the final parameters.  The
    re-packed weight fragments are what the second and later iterations compute with: a wrong fragment shows in their
    losses."""
    from prob_mbrl_amd import engine as E
    d = np.load(common.os.path.join(common.GOLDEN, name + '.npz'))
    dev = torch.device('cuda:0')
    nl = int(d['n_layers'])
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)  # noqa: E731
    parts, keys = [], []
    for l in range(nl):
        parts += [T(d['W%d_init' % l]).reshape(-1), T(d['b%d_init' % l]).reshape(-1)]
        keys += ['W%d' % l, 'b%d' % l]
        if l < nl - 1:
            parts.append(T(d['logit_p%d_init' % l]).reshape(-1))
            keys.append('logit_p%d' % l)
    flat = torch.cat(parts).contiguous()
    sizes = [p.numel() for p in parts]
    dims = [d['W0_init'].shape[1]] + [d['W%d_init' % l].shape[0] for l in range(nl)]
    M, N, iters = int(d['M']), int(d['N']), int(d['iters'])
    step = E.BnnStep(dims, [float(d['temp%d' % l]) for l in range(nl - 1)], [float(d['reg_scale%d' % l]) for l in range(nl - 1)],
                     [float(d['drop_reg%d' % l]) for l in range(nl - 1)], M, N, float(d['reg_weight']),
                     max_log_std=float(d['max_log_std']), device=dev)
    Xn, Yn = T(d['Xn']), T(d['Yn'])
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    idx = torch.tensor(np.stack([d['idx_it%d' % it].astype(np.int32) for it in range(iters)]), device=dev).contiguous()
    u = torch.stack([torch.cat([T(d['u%d_it%d' % (l, it)]).reshape(-1) for l in range(nl - 1)]) for it in range(iters)]).contiguous()
    bvar = torch.stack([torch.cat([1.0 - T(d['hard%d_it%d' % (l, it)]).reshape(-1) for l in range(nl - 1)])
                        for it in range(iters)]).contiguous()
    stepc = torch.zeros(1, dtype=torch.int64, device=dev)
    hist = torch.zeros(iters, 3, device=dev)
    loss = step.train_steps(Xn, Yn, idx, flat, m, v, stepc, float(d['lr']), u=u, bvar=bvar, loss_hist=hist)
    torch.cuda.synchronize()
    assert int(stepc.item()) == iters
    assert np.allclose(hist.cpu().numpy(), d['losses'][:iters], rtol=3e-5, atol=2e-6), (hist, d['losses'])
    assert torch.equal(loss.cpu(), hist[-1].cpu())
    off = 0
    for key, n in zip(keys, sizes):
        got = flat[off:off + n].cpu().numpy()
        want = d[key + '_final'].reshape(-1)
        assert np.allclose(got, want, rtol=1e-4, atol=2e-6), (key, np.abs(got - want).max())
        off += n
    # the in-kernel noise: reproducible from its seed, different from step to step, and the loss of a fit goes down
    flat2, m2, v2 = torch.cat(parts).contiguous(), torch.zeros_like(flat), torch.zeros_like(flat)
    runs = []
    for _ in range(2):
        f, mm, vv, sc = flat2.clone(), m2.clone(), v2.clone(), torch.zeros(1, dtype=torch.int64, device=dev)
        h = torch.zeros(40, 3, device=dev)
        ix = torch.randint(0, N, (40, M), device=dev, dtype=torch.int32, generator=torch.Generator(device=dev).manual_seed(3))
        step.train_steps(Xn, Yn, ix, f, mm, vv, sc, 1e-3, seed=1234, first_step=0, loss_hist=h)
        runs.append((f.cpu(), h.cpu()))
    assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
    assert bool(torch.isfinite(runs[0][1]).all()) and float(runs[0][1][-5:, 0].mean()) < float(runs[0][1][:5, 0].mean())


@pytest.mark.parametrize('name', common.fixture_names('bnn'))
def test_bnn_training_matches_reference(name):
    """pmbrl_bnn_loss_grad + pmbrl_clip_adam replaying the reference's train_regressor
    iterations (recorded minibatch indices, concrete-dropout noise and Bernoulli draws):
    losses, first-iteration gradients and final parameters."""
    from prob_mbrl_amd import engine as E
    d = np.load(common.os.path.join(common.GOLDEN, name + '.npz'))
    dev = torch.device('cuda:0')
    nl = int(d['n_layers'])
    T = lambda a: torch.tensor(np.asarray(a, dtype=np.float32), device=dev)  # noqa: E731
    parts, keys = [], []
    for l in range(nl):
        parts += [T(d['W%d_init' % l]).reshape(-1), T(d['b%d_init' % l]).reshape(-1)]
        keys += ['W%d' % l, 'b%d' % l]
        if l < nl - 1:
            parts.append(T(d['logit_p%d_init' % l]).reshape(-1))
            keys.append('logit_p%d' % l)
    flat = torch.cat(parts).contiguous()
    sizes = [p.numel() for p in parts]
    dims = [d['W0_init'].shape[1]] + [d['W%d_init' % l].shape[0] for l in range(nl)]
    temps = [float(d['temp%d' % l]) for l in range(nl - 1)]
    rs = [float(d['reg_scale%d' % l]) for l in range(nl - 1)]
    dr = [float(d['drop_reg%d' % l]) for l in range(nl - 1)]
    M, N = int(d['M']), int(d['N'])
    n_comp = int(d['n_components']) if 'n_components' in d else 0      # > 1: mixture-of-Gaussians head + NLL
    step = E.BnnStep(dims, temps, rs, dr, M, N, float(d['reg_weight']), max_log_std=float(d['max_log_std']),
                     device=dev, loss_kind='gmm' if n_comp > 1 else 'nll', n_components=n_comp)
    assert step.n_params == flat.numel()
    Xn, Yn = T(d['Xn']), T(d['Yn'])
    m, v = torch.zeros_like(flat), torch.zeros_like(flat)
    for it in range(int(d['iters'])):
        idx = torch.tensor(d['idx_it%d' % it].astype(np.int32), device=dev)
        u = torch.cat([T(d['u%d_it%d' % (l, it)]).reshape(-1) for l in range(nl - 1)])
        hard = torch.cat([T(d['hard%d_it%d' % (l, it)]).reshape(-1) for l in range(nl - 1)])
        bvar = 1.0 - hard                      # bvar < probs  <=>  hard  (probs in (0, 1))
        grad, loss = step.loss_grad(Xn, Yn, idx, flat, u.contiguous(), bvar.contiguous())
        assert np.allclose(loss.cpu().numpy(), d['losses'][it], rtol=3e-5, atol=2e-6), (it, loss, d['losses'][it])
        if it == 0:
            off = 0
            for key, n in zip(keys, sizes):
                got = grad[off:off + n].cpu().numpy()
                assert common.rel(got, d['g%s_it0' % key].reshape(-1)) < 5e-5, key
                off += n
        E.clip_adam(flat, grad, m, v, it + 1, float(d['lr']), max_norm=None)
    off = 0
    for key, n in zip(keys, sizes):
        got = flat[off:off + n].cpu().numpy()
        assert np.allclose(got, d[key + '_final'].reshape(-1), rtol=2e-5, atol=5e-7), key
        off += n


@pytest.mark.gpu
def test_mlp_input_gradient_matches_torch():
    """pmbrl_mlp_grad_input (autograd through a stand-alone network) against torch autograd on
    the same formula."""
    from prob_mbrl_amd import engine as E
    g = torch.Generator().manual_seed(5)
    dev = torch.device('cuda:0')
    B, dims = 45, [6, 40, 56, 6]
    nl, n_out = 3, 3
    Ws = [torch.randn(dims[i + 1], dims[i], generator=g) / np.sqrt(dims[i]) for i in range(nl)]
    bs = [0.1 * torch.randn(dims[i + 1], generator=g) for i in range(nl)]
    masks = [(torch.rand(B, dims[i + 1], generator=g) < 0.8).float() for i in range(nl - 1)]
    keep = [0.8, 1.0]
    x = torch.randn(B, dims[0], generator=g, dtype=torch.float64).requires_grad_(True)
    z = torch.randn(B, n_out, generator=g)
    shift, iscale = torch.randn(dims[0], generator=g), 0.5 + torch.rand(dims[0], generator=g)
    osc, osh = 0.5 + torch.rand(n_out, generator=g), torch.randn(n_out, generator=g)
    sqs, sqb = 1.0 + torch.rand(n_out, generator=g), 0.1 * torch.randn(n_out, generator=g)
    cs, cm, cl = (torch.randn(B, n_out, generator=g) for _ in range(3))
    d = lambda t: t.double()  # noqa: E731
    h = (x - d(shift)) * d(iscale)
    for i in range(nl - 1):
        h = torch.relu(h @ d(Ws[i]).t() + d(bs[i])) * d(masks[i]) / keep[i]
    o = h @ d(Ws[-1]).t() + d(bs[-1])
    mu, ls = o[:, :n_out], o[:, n_out:]
    mls = float(np.log(5.0))
    ls = -torch.nn.functional.softplus(-ls + mls) + mls + d(osc).log()
    mu = mu * d(osc) + d(osh)
    smp = d(sqs) * torch.tanh(mu + d(z) * ls.exp()) + d(sqb)
    ((smp * d(cs)).sum() + (mu * d(cm)).sum() + (ls * d(cl)).sum()).backward()
    flat = torch.cat([t.reshape(-1) for pair in zip(Ws, bs) for t in pair]).to(dev)
    bits = [E.pack_mask(m.to(dev)) for m in masks]
    xg = x.detach().float().to(dev).requires_grad_(True)
    out = E.mlp_forward(xg, flat, dims, keep, bits, z.to(dev), shift, iscale, osc, osh, sqs, sqb,
                        max_log_std=mls, want=('sample', 'mean', 'log_std'))
    ((out['sample'] * cs.to(dev)).sum() + (out['mean'] * cm.to(dev)).sum() +
     (out['log_std'] * cl.to(dev)).sum()).backward()
    assert common.rel(xg.grad.cpu().numpy(), x.grad.numpy()) < 2e-5


@pytest.mark.gpu
def test_invalid_plans_fail_cleanly():
    """Bad configurations are usage errors with a message (negative status through the C ABI ->
    RuntimeError on the Python side), never a crash or a silent fallback."""
    from prob_mbrl_amd import engine as E
    d = common.load('nomm_d4')
    from prob_mbrl_amd import problem as PB
    spec = PB.cartpole_reward_spec(4)
    eng, args, _ = common.engine_from_fixture(d, torch.device('cuda:0'))
    good = dict(B=eng.B, D=eng.D, U=eng.U, H=eng.H)
    pol = [eng.D, 32, 32, 2 * eng.U]
    dyn = [eng.D + eng.U, 32, 32, 2 * eng.D]

    def make(**kw):
        a = dict(good)
        a.update(kw)
        return E.Engine(a['B'], a['D'], a['U'], a['H'], kw.get('pol', pol), [0.9, 0.9], kw.get('dyn', dyn),
                        [1.0, 1.0], spec, mm_states=kw.get('mm', False), mm_rewards=kw.get('mm', False),
                        mm_groups=kw.get('groups'), device='cuda:0')

    for bad in (dict(B=0), dict(H=0), dict(D=40, pol=[40, 32, 2], dyn=[41, 32, 80]),
                dict(mm=True, groups=7),                      # B not divisible by the groups
                dict(mm=True, groups=good['B'])):             # one row per group
        from prob_mbrl_amd._lib import PmbrlError
        with pytest.raises((PmbrlError, ValueError, AssertionError)):
            make(**bad)
    # a forward call with a missing input is refused, and the plan stays usable afterwards
    broken = dict(args)
    broken['z_pol'] = None
    with pytest.raises((PmbrlError, AssertionError, AttributeError, TypeError)):
        eng.forward(**broken)
    S, A, R = eng.forward(**args)
    assert torch.isfinite(S).all() and eng.valid_steps() == eng.H


def test_rccl_allreduce_through_the_c_abi(dev):
    """pmbrl_comm_* / pmbrl_allreduce_sum: the RCCL communicator behind the C ABI, bootstrapped and
    exercised with the one rank a single-GPU box has (the all-reduce of one rank is the identity; what
    this pins is the dlopen of librccl, the symbol signatures and the call on torch's stream).  The
    multi-rank run is the driver's scaling bench."""
    import ctypes as C
    from prob_mbrl_amd import _lib, engine as E
    lib = _lib.load()
    idbuf = C.create_string_buffer(128)
    _lib.check(lib.pmbrl_comm_unique_id(idbuf), 'pmbrl_comm_unique_id')
    assert any(idbuf.raw)
    comm = C.c_void_p()
    _lib.check(lib.pmbrl_comm_init(C.c_char_p(bytes(idbuf.raw)), 0, 1, 0, C.byref(comm)), 'pmbrl_comm_init')
    g = torch.randn(41602, device=dev)
    want = g.clone()
    for _ in range(3):
        _lib.check(lib.pmbrl_allreduce_sum(comm, E._stream(), E._ptr(g), g.numel()), 'pmbrl_allreduce_sum')
    torch.cuda.synchronize()
    assert torch.equal(g, want)
    lib.pmbrl_comm_destroy(comm)


# ---------------------------------------------------------------------------
# dropout masks drawn on the device (pmbrl_draw_masks)
# ---------------------------------------------------------------------------
def _unpack(bits, h):
    b = bits.cpu().numpy().view(np.uint16)
    out = np.unpackbits(b.view(np.uint8).reshape(b.shape[0], -1), axis=1, bitorder='little')
    return out[:, :h]


def test_draw_masks_formula_on_the_references_recorded_uniforms(dev):
    """Fixture draw_dropout: the reference's own CDropout.update_noise / BDropout.update_noise calls (uniforms u, the
    probabilities it handed to torch.bernoulli, the hard samples it kept).  With its u the kernel must form the same
    probabilities (models/modules.py:102-114), and with threshold uniforms on the recorded side of them, the same bits."""
    import os
    from prob_mbrl_amd import engine as E
    d = np.load(os.path.join(common.GOLDEN, 'draw_dropout.npz'))
    B, h = int(d['B']), int(d['h'])
    for k in range(2):
        u = torch.tensor(d['u%d' % k], device=dev).contiguous()
        probs, hard = d['probs%d' % k].astype(np.float64), d['hard%d' % k].astype(bool)
        v = torch.tensor(np.where(hard, 0.5 * probs, 0.5 * (1.0 + probs)).astype(np.float32), device=dev)
        bits, aux = E.draw_masks('concrete', 0, 0, torch.tensor(d['logit_p%d' % k], device=dev), float(d['temp%d' % k]),
                                 B, h, u=u, v=v, aux=(0, B), want=('u', 'hard', 'probs'))
        assert np.allclose(aux['probs'].cpu().numpy(), probs, rtol=2e-5, atol=1e-7)
        assert np.array_equal(_unpack(bits, h).astype(bool), hard)
        assert np.array_equal(aux['hard'].cpu().numpy().astype(bool), hard) and torch.equal(aux['u'], u)
    keep = 1.0 - d['b_rate'].astype(np.float64)
    hb = d['hardb'].astype(bool)
    u = torch.tensor(np.where(hb, 0.5 * keep, 0.5 * (1.0 + keep)).astype(np.float32), device=dev).expand(B, h).contiguous()
    bits, _ = E.draw_masks('bernoulli', 0, 0, torch.tensor(keep.astype(np.float32), device=dev), 0.0, B, h, u=u)
    assert np.array_equal(_unpack(bits, h).astype(bool), hb)


def test_draw_masks_distribution_and_reproducibility(dev):
    """The generator's own draws: per-unit frequencies against the probabilities (Bernoulli: keep; concrete: the mean
    of torch's own draw of the same formula), 4 sigma; no correlation between neighbouring units or rows; a draw is
    a function of (seed, offset) -- same again, different with another offset, independent of how many rows are asked."""
    from prob_mbrl_amd import engine as E
    rows, h = 8192, 200
    keep = torch.tensor([0.9], device=dev)
    b0, _ = E.draw_masks('bernoulli', 1234, 0, keep, 0.0, rows, h)
    b1, _ = E.draw_masks('bernoulli', 1234, 0, keep, 0.0, rows, h)
    b2, _ = E.draw_masks('bernoulli', 1234, 1, keep, 0.0, rows, h)
    b3, _ = E.draw_masks('bernoulli', 1234, 0, keep, 0.0, 100, h)
    assert torch.equal(b0, b1) and not torch.equal(b0, b2) and torch.equal(b0[:100], b3)
    m = _unpack(b0, h).astype(np.float64)
    sig = np.sqrt(0.9 * 0.1 / rows)
    assert np.all(np.abs(m.mean(0) - 0.9) < 4.5 * sig) and abs(m.mean() - 0.9) < 4 * sig / np.sqrt(h)
    c = m - m.mean(0)
    assert abs((c[:, :-1] * c[:, 1:]).mean()) < 4 * 0.09 / np.sqrt(rows * (h - 1))      # neighbouring units
    assert abs((c[:-1] * c[1:]).mean()) < 4 * 0.09 / np.sqrt((rows - 1) * h)             # neighbouring rows
    # concrete dropout: per-unit logits, temperature 0.1
    g = torch.Generator().manual_seed(5)
    logit = (torch.randn(h, generator=g) * 1.5 + 1.0).to(dev)
    bits, aux = E.draw_masks('concrete', 77, 3, logit, 0.1, rows, h, aux=(0, rows), want=('u', 'hard', 'probs'))
    hard = _unpack(bits, h).astype(np.float64)
    assert np.array_equal(hard, aux['hard'].cpu().numpy())
    u = aux['u']
    assert float(u.min()) >= 0.0 and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 4 / np.sqrt(12.0 * rows * h)
    probs_t = (((logit + ((u + 1e-7) / (1 - (u - 1e-7))).log()) / 0.1).sigmoid())
    assert torch.allclose(aux['probs'], probs_t, rtol=2e-5, atol=1e-7)
    p_unit = probs_t.double().mean(0).cpu().numpy()
    sd = np.sqrt(np.maximum(p_unit * (1 - p_unit), 1e-4) / rows)
    assert np.all(np.abs(hard.mean(0) - p_unit) < 5 * sd)
    # ... and torch's own draw of the same thing has the same per-unit frequencies
    ut = torch.rand(rows, h, device=dev)
    ht = torch.bernoulli(((logit + ((ut + 1e-7) / (1 - (ut - 1e-7))).log()) / 0.1).sigmoid()).double().mean(0).cpu().numpy()
    assert np.all(np.abs(hard.mean(0) - ht) < 7 * sd)


def test_rollout_with_per_step_masks_uses_the_device_draw(dev):
    """utils.rollout(resample_model=True, resample_policy=True): one pmbrl_draw_masks launch per dropout layer; masks
    differ from step to step, the run is reproducible under torch.manual_seed, and the concrete-dropout modules are
    left holding the last step's hard sample next to their untouched stored noise (models/modules.py:134-143,155-157)."""
    import prob_mbrl_amd as pm
    d = common.load('nomm_d4')
    dyn, pol = common.modules_from_fixture(d, 'nomm_d4', 'cuda:0')
    x0 = torch.tensor(d['x0'], device=dev)
    H, B = int(d['H']), x0.shape[0]
    kw = dict(resample_model=True, resample_policy=True, resample_state_noise=False, resample_action_noise=False)
    torch.manual_seed(11)
    S1, _, _ = pm.utils.rollout(x0, dyn, pol, H, **kw)
    n1, c1 = dyn.model.drop0.noise.clone(), dyn.model.drop0.concrete_noise.clone()
    torch.manual_seed(11)
    S2, _, _ = pm.utils.rollout(x0, dyn, pol, H, **kw)
    S3, _, _ = pm.utils.rollout(x0, dyn, pol, H, **kw)
    assert all(torch.equal(a, b) for a, b in zip(S1, S2)) and not torch.equal(S1[-1], S3[-1])
    assert torch.isfinite(torch.stack(S3)).all()
    assert n1.shape == (B, dyn.model.drop0.logit_p.numel()) and set(np.unique(c1.cpu().numpy())) <= {0.0, 1.0}
    # forward(resample=True) draws its uniforms into a local: the stored noise stays, the hard sample is the last step's
    assert torch.equal(n1, dyn.model.drop0.noise)
    assert c1.shape == dyn.model.drop0.concrete_noise.shape and not torch.equal(c1, dyn.model.drop0.concrete_noise)


# ---------------------------------------------------------------------------
# fused iteration tail (pmbrl_plan_set_loss, pmbrl_rollout_bwd_adam)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize('name,generic', [('nomm_d4', False), ('mmg_d4', False), ('full200_nomm', False), ('nomm_d5', True)])
def test_fused_tail_matches_the_separate_launches(dev, name, generic):
    """The iteration tail queued by two calls (pmbrl_plan_set_loss: the loss behind the forward call;
    pmbrl_rollout_bwd_adam: adjoint, dW, clip and the device-guarded Adam) against the separate calls (weighted_sum,
    backward, clip_adam): the same loss and clipped gradient, parameters and moments to rounding, over three iterations;
    a rollout marked as failed leaves parameters, moments and the step counter alone.  (Each iteration is replayed from
    the fused path's own parameters -- see below; the comparison of the two paths' OWN three-iteration trajectories is
    tests/test_gpu_full_size.py::test_fused_tail_walks_the_separate_launches_trajectory_at_full_size, at 2 500 rows.)"""
    from prob_mbrl_amd import engine as E
    d = common.load(name)
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)

    # The fused calls walk three iterations; every one of them is then replayed through the separate calls FROM THE SAME
    # parameters and moments (not from the separate calls' own trajectory: the device-side step takes its bias corrections
    # from device pow(), equal to the host's to rounding, and a policy gradient is not continuous in the parameters -- with
    # 37 rows one ReLU unit changing sides under a 1e-7 parameter difference moves it by 1e-3, which says nothing about
    # the fused tail).  Same parameters: the same kernels, the same bits for loss and gradient; the update to rounding.
    eng, args, _ = common.engine_from_fixture(d, dev, force_generic=generic)
    p = args['pol_flat'].clone()
    args['pol_flat'] = p
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    step = torch.zeros(1, dtype=torch.int64, device=dev)
    norm = torch.zeros(1, device=dev)
    loss_buf = eng.set_loss(gw)
    walk = []
    for it in range(1, 4):
        before = (p.clone(), m.clone(), v.clone())
        eng.forward(**args)
        g, _, _ = eng.backward(gw, adam=dict(params=p, exp_avg=m, exp_avg_sq=v, step=step, lr=1e-3,
                                             betas=(0.9, 0.999), eps=1e-8, max_norm=0.05, norm_out=norm))
        walk.append((before, float(loss_buf), float(norm), g.clone(), p.clone(), m.clone(), v.clone()))
    assert int(step.item()) == 3
    ref_eng, ref_args, _ = common.engine_from_fixture(d, dev, force_generic=generic)
    for it, (before, loss_f, norm_f, g_f, p_f, m_f, v_f) in enumerate(walk, 1):
        pr, mr, vr = (t.clone() for t in before)
        ref_args['pol_flat'] = pr
        ref_loss, ref_norm = torch.zeros(1, device=dev), torch.zeros(1, device=dev)
        _, _, R = ref_eng.forward(**ref_args)
        ref_eng.weighted_sum(R, gw, out=ref_loss)
        g_r, _, _ = ref_eng.backward(gw)
        g_r = g_r.clone()
        E.clip_adam(pr, g_r, mr, vr, it, 1e-3, max_norm=0.05, norm_out=ref_norm)
        assert float(ref_loss) == loss_f and torch.equal(g_r, g_f)
        assert abs(float(ref_norm) - norm_f) <= 1e-6 * norm_f
        for x, y in ((pr, p_f), (mr, m_f), (vr, v_f)):
            assert torch.allclose(x, y, rtol=1e-4, atol=1e-5 * float(x.abs().max()))
    # a failed rollout: the status word says step 3 of H failed -> nothing moves
    eng.forward(**args)
    eng.status[0] = 3
    before = (p.clone(), m.clone(), v.clone())
    eng.backward(gw, adam=dict(params=p, exp_avg=m, exp_avg_sq=v, step=step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8,
                               max_norm=0.05))
    assert int(step.item()) == 3 and all(torch.equal(x, y) for x, y in zip(before, (p, m, v)))


@pytest.mark.parametrize('name', ['c5_small', 'c5_mm_small'])
def test_in_place_layers_at_64_rows_per_workgroup(dev, name):
    """General family on split operands with rows_per_wg_hint = 64 at 3 x 512: ONE activation buffer, every layer
    written in place after the workgroup has read it (pmbrl_gsplit.h: gemm_layer_inplace_s; the two-buffer form does
    not fit 64 rows in LDS).  Same fixtures from the reference, same tolerances, and agreement with the 16-row form."""
    d = common.load(name)
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    out = []
    for hint in (0, 64):
        eng, args, _ = common.engine_from_fixture(d, dev, rows_per_wg_hint=hint, force_generic=True)
        S, A, Rw = eng.forward(**args)
        g = eng.backward(gw)[0].cpu().numpy().copy()
        assert eng.valid_steps() == int(d['H'])
        out.append((eng.info['rows_per_wg'], eng.info['lds_bytes'], S.cpu().numpy(), g))
    assert out[0][0] == 16 and out[1][0] == 64 and out[1][1] < 160 * 1024
    for _, _, S, g in out:
        assert common.rel(S, d['ref64_states']) < TOL_TRAJ and common.rel(g, d['ref64_grad']) < TOL_GRAD
    assert common.rel(out[1][2], out[0][2]) < 1e-6 and common.rel(out[1][3], out[0][3]) < 1e-5


@pytest.mark.parametrize('name', ['full200_nomm', 'full200_mmg'])
def test_lds_resident_tiles_match_the_streamed_form(dev, name, monkeypatch):
    """The shape-specialised 16-row instances keep 8 weight tiles of the second streamed layer in LDS (pmbrl_fast.h:
    lds_tile_s) -- plain variants and in-kernel moment matching; PMBRL_LDS_TILES=0 runs the same plan on the generic
    instances, which stream every tile: same fixtures, same tolerances, and the two agree."""
    d = common.load(name)
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    out = []
    for off in (False, True):
        if off:
            monkeypatch.setenv('PMBRL_LDS_TILES', '0')
        eng, args, _ = common.engine_from_fixture(d, dev)
        S, A, Rw = eng.forward(**args)
        g = eng.backward(gw)[0].cpu().numpy().copy()
        assert eng.valid_steps() == int(d['H']) and eng.info['fast'] and eng.info['rows_per_wg'] <= 16
        out.append((eng.info['lds_bytes'], S.cpu().numpy(), g))
    monkeypatch.delenv('PMBRL_LDS_TILES')
    assert out[0][0] > out[1][0] + 100 * 1024          # (8 tiles of 13 KB)
    for _, S, g in out:
        assert common.rel(S, d['ref64_states']) < TOL_TRAJ and common.rel(g, d['ref64_grad']) < TOL_GRAD
    assert common.rel(out[1][1], out[0][1]) < 2e-6 and common.rel(out[1][2], out[0][2]) < 2e-5
