"""GPU (-m gpu): BASELINE.json's FULL-SIZE configurations (2500 x 40 cart-pole rows with and without
in-kernel moment matching, the 50-row-group double cart-pole shape).  These are the shapes the
shape-specialised LEAN / MM instantiations serve (what bench.py and mc_pilco's fused iteration
launch), which the small fixtures never reach.  Checked against the oracle where it finishes in
seconds, and through size-independent properties everywhere: the specialised, the general and
the EXT instantiations are the same arithmetic (bit-identical results), the gradient is linear in
the loss weights, and rows are independent (a row permutation permutes the trajectories)."""
import os
import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _errors_vs_oracle(d, **kw):
    from oracle import ref_torch as R
    eng, S, A, Rw, loss, g, _ = _run(d, **kw)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                            meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
    return eng, common.rel(S, torch.stack(S64).detach().numpy()), abs(loss - float(l64)) / abs(float(l64)), \
        common.rel(g, g64.numpy())


@pytest.mark.parametrize('prec', ['split', 'split_f16'])
@pytest.mark.parametrize('config', ['cartpole_nomm', 'cartpole_mm'])
def test_full_size_split_precision_matches_oracle(config, prec):
    """C2 / C3 at full size on the split-operand matrix-core paths against the fp64 oracle, same
    tolerances as the exact-fp32 path."""
    d = _problem(config)
    eng, e_s, e_l, e_g = _errors_vs_oracle(d, precision=prec)
    assert eng.info['precision'] == prec and eng.info['fast'] == 1
    if prec == 'split_f16':
        # the headline kernels (pm_reg_fwd_kernel / pm_reg_bwd_kernel) served BOTH sweeps: a plan the family silently
        # declined would otherwise pass here on the eight-wave kernels
        assert eng.info['reg'] and eng.reg_calls() == (1, 1), (eng.info, eng.reg_calls())
    print('%s %s: states %.2e loss %.2e grad %.2e' % (config, prec, e_s, e_l, e_g))
    assert e_s < 2e-5 and e_l < 2e-5 and e_g < 1e-4


def _run(d, lean=True, **kw):
    from prob_mbrl_amd import problem as PB
    eng, args, _ = PB.engine_from_problem(d, DEV, **kw)
    B = d['x0'].shape[0]
    gw = torch.tensor(PB.loss_weights(d, B), device=DEV)
    S, A, R = eng.forward(**args)
    loss = float(eng.weighted_sum(R, gw))
    if lean:
        g, _, _ = eng.backward(gw)
    else:   # optional outputs select the EXT instantiation
        g, _, _ = eng.backward(gw, want_x0=True, want_agn=True)
    assert eng.valid_steps() == int(d['H'])
    return eng, S.cpu().numpy(), A.cpu().numpy(), R.cpu().numpy(), loss, g.cpu().numpy().copy(), gw


def _problem(config, H=None):
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem(config, seed=0, data_seed=0))
    if H is not None:
        d['H'] = np.asarray(H)
        d['gamma'] = np.asarray(d['gamma'])[:H] * (len(np.asarray(d['gamma'])) / H)
    return d


@pytest.mark.parametrize('config', ['cartpole_nomm', 'cartpole_mm'])
def test_full_size_matches_oracle(config):
    """C2 / C3 at full size against the torch-CPU restatement of the reference in fp64."""
    from oracle import ref_torch as R
    d = _problem(config)
    eng, S, A, Rw, loss, g, _ = _run(d)
    # (cartpole_mm: every 25-row group split over two 16-row workgroups, 13 + 12 rows)
    assert eng.info['fast'] == 1 and eng.info['rows_per_wg'] == (16 if config == 'cartpole_nomm' else 13)
    assert eng.info['mm_parts'] == (1 if config == 'cartpole_nomm' else 2)
    # ... and on the register-resident family, both sweeps (the kernels bench.py's headline line times)
    assert eng.info['reg'] and eng.reg_calls() == (1, 1), (eng.info, eng.reg_calls())
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(8)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                            meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert common.rel(A, torch.stack(A64).detach().numpy()) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4


@pytest.mark.parametrize('config,H', [('cartpole_nomm', None), ('cartpole_mm', None), ('dcartpole_mm', 12)])
def test_instantiations_agree_bitwise(config, H):
    """Shape-specialised vs general instantiation, LEAN vs EXT variant: identical arithmetic."""
    d = _problem(config, H)
    # (64-row workgroups on split operands exist as the shape-specialised instance only -- pmbrl.hip,
    #  rt4_split -- so the double cart-pole comparison is made in fp32, where both forms exist)
    pk = dict(precision='f32') if config == 'dcartpole_mm' else {}
    ref = _run(d, **pk)
    # cartpole_mm: the shape-specialised instance exchanges fp64 SUMS between the two workgroups of a 25-row group
    # (pm_xch_put / pm_xch_get), the general instance carries the rows + flags form -- the same mathematics in a
    # different order of fp64 additions, so their fp32 results agree to rounding, not to the bit (identical bits
    # there were a matter of which way ~1e-14 differences round).  cartpole_nomm: the plain whole-horizon sweeps run
    # on the register-resident family (pmbrl_reg.h: another K order inside a hidden-width product), the general and
    # the EXT instances on the latency-optimised one.  Everything else is the same arithmetic in the same order and
    # must agree bit for bit.
    reg = bool(ref[0].info.get('reg'))
    # (round 5: the register-resident family also serves the moment-matching cart-pole shapes -- pmbrl_reg_mm.h)
    assert reg == (config in ('cartpole_nomm', 'cartpole_mm'))
    outs = {}
    for kw, lean in ((dict(no_shaped=True), True), (dict(), False), (dict(no_shaped=True), False)):
        outs[(bool(kw), lean)] = _run(d, lean=lean, **kw, **pk)

    def same_bits(a, b):
        assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[3], b[3])
        assert np.array_equal(a[5], b[5])

    def same_rounding(a, b, g_tol=2e-5):
        assert common.rel(b[1], a[1]) < 2e-6 and common.rel(b[2], a[2]) < 2e-6 and common.rel(b[5], a[5]) < g_tol, \
            (common.rel(b[1], a[1]), common.rel(b[2], a[2]), common.rel(b[5], a[5]))

    if config == 'cartpole_mm':
        assert ref[0].info['mm_parts'] == 2
        # ref: both sweeps on the register-resident family; (False, False): its forward, the latency-optimised family's
        # adjoint from the same stashes; the general instances carry the rows + flags form of the split groups -- the same
        # mathematics in another order of fp64 additions
        for k in (1, 2, 3):
            assert np.array_equal(ref[k], outs[(False, False)][k])
        # (the two families' adjoints of the moment matching differ in more than the order of a sum -- triangular solves
        #  in scalar fp64 there, products with the stashed L^-1 on the fp64 matrix core here -- and the adjoint of a
        #  Cholesky factor amplifies what differs: against the fp64 oracle they measure 1.6e-5 and 3.1e-5
        #  (test_full_size_matches_oracle holds each to 1e-4), against each other 2.9e-5)
        same_rounding(ref, outs[(False, False)], g_tol=6e-5)
        same_bits(outs[(True, True)], outs[(True, False)])
        same_rounding(ref, outs[(True, True)], g_tol=6e-5)
    elif reg:
        # (the EXT call's forward is the same register-resident launch: identical trajectories; its adjoint is the
        #  latency-optimised family's)
        for k in (1, 2, 3):
            assert np.array_equal(ref[k], outs[(False, False)][k])
        same_rounding(ref, outs[(False, False)])
        same_bits(outs[(True, True)], outs[(True, False)])
        same_rounding(ref, outs[(True, True)])
    else:
        for o in outs.values():
            same_bits(ref, o)


def test_gradient_is_linear_in_loss_weights():
    d = _problem('cartpole_nomm')
    eng, S, A, Rw, loss, g, gw = _run(d)
    g2, _, _ = eng.backward(2.0 * gw)
    assert common.rel(g2.cpu().numpy(), 2.0 * g) < 1e-6        # (not bit-exact: denormals are flushed)
    w1 = gw * torch.rand_like(gw)
    ga = eng.backward(w1)[0].cpu().numpy().copy()
    gb = eng.backward(gw - w1)[0].cpu().numpy().copy()
    # (two bf16 pieces per operand in every product of the adjoint sweep: 2^-17 per product, the sum of two roundings
    #  is not the rounding of the sum)
    assert common.rel(ga + gb, g) < 4e-6


def test_rows_are_independent():
    """Without moment matching a row's trajectory depends on nothing but its own inputs: reversing
    the row order reverses the trajectories bit for bit (different workgroups, different tile
    positions) and leaves the gradient unchanged up to summation order."""
    d = _problem('cartpole_nomm')
    ref = _run(d)
    e = dict(d)
    for k in list(e):
        v = np.asarray(e[k])
        if k == 'x0' or k in ('pol_z', 'dyn_z') or ('_mask' in k and v.ndim == 2 and v.shape[0] == d['x0'].shape[0]):
            e[k] = v[::-1].copy()
    out = _run(e)
    assert np.array_equal(ref[1][:, ::-1], out[1]) and np.array_equal(ref[2][:, ::-1], out[2])
    assert common.rel(out[5], ref[5]) < 2e-6


@pytest.mark.parametrize('groups', [25, 0], ids=['100-row groups', 'one 2500-row group'])
def test_groups_spanning_workgroups_match_oracle(groups):
    """Moment-matching groups larger than a workgroup's 16 rows (mm_mode 3: the statistics are
    recomputed by every workgroup after a device-wide barrier, or in the prologue of per-step launches).  100-row groups straddle
    workgroup boundaries (100 is not a multiple of 16); mm_groups=None is the reference examples'
    default.  Against the fp64 oracle, and against the separate-kernel path (mm_mode 2)."""
    import os
    from oracle import ref_torch as R
    d = _problem('cartpole_mm', 12)
    d['mm_groups'] = np.asarray(groups)
    # (100-row groups would by default be split over seven workgroups with a group-local barrier -- below; this
    #  test is about the device-wide-barrier form, which serves the groups of more than 128 rows)
    if groups:
        eng7, S7, A7, Rw7, loss7, g7, _ = _run(d)
        assert eng7.info['mm_mode'] == 1 and eng7.info['mm_parts'] == 7 and eng7.info['rows_per_wg'] == 15
    os.environ['PMBRL_MM_PARTS'] = '1'
    try:
        _spanning_forms(d, groups, (S7, g7) if groups else None)
    finally:
        del os.environ['PMBRL_MM_PARTS']


@pytest.mark.parametrize('groups', [0, 5], ids=['one 2500-row group', '500-row groups'])
def test_large_groups_split_over_many_workgroups_match_oracle(groups):
    """Groups beyond 8 x 32 rows on the cart-pole shape: 16-row parts whose Gram sums travel over two levels (collectors
    of ~sqrt(parts) parts, pm_xch_get_tree) instead of the device-wide barrier form -- mm_groups=None, the reference
    examples' default (examples/deep_pilco_mm.py:31), is 157 parts.  Against the fp64 oracle at the plain bars, and
    against the device-wide barrier form (other summation order: rounding only)."""
    import os
    from oracle import ref_torch as R
    d = _problem('cartpole_mm', 12)
    d['mm_groups'] = np.asarray(groups)
    eng, S, A, Rw, loss, g, _ = _run(d)
    M = 2500 // max(groups, 1)
    assert eng.info['mm_mode'] == 1 and eng.info['mm_parts'] == (M + 15) // 16 and eng.info['rows_per_wg'] == 16
    assert eng.valid_steps() == int(d['H'])
    # (round 6: both sweeps on the register-resident family -- pmbrl_reg_mm.h with the two-level exchange)
    assert eng.info['reg'] and eng.reg_calls() == (1, 1), (eng.info, eng.reg_calls())
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(8)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True,
                                            meta['mm_groups'], z_mm, z_rr)
    print('groups %d: states %.2e loss %.2e grad %.2e' % (groups, common.rel(S, torch.stack(S64).detach().numpy()),
                                                           abs(loss - float(l64)) / abs(float(l64)), common.rel(g, g64.numpy())))
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4
    os.environ['PMBRL_MM_TREE'] = '0'
    try:
        eng3, S3, A3, Rw3, loss3, g3, _ = _run(d)
    finally:
        del os.environ['PMBRL_MM_TREE']
    assert eng3.info['mm_mode'] in (2, 3)
    print('   vs the device-wide barrier form: states %.2e grad %.2e' % (common.rel(S3, S), common.rel(g3, g)))
    # (the register-resident family's adjoint of the moment matching is products with the stashed L^-1 on the fp64 matrix
    #  core where the barrier form solves triangular systems in scalar fp64: the bar of _spanning_forms)
    assert common.rel(S3, S) < 5e-6 and common.rel(g3, g) < 6e-5
    # the same exchange on the latency-optimised family (what serves a call with optional outputs on this plan)
    os.environ['PMBRL_REG_TREE'] = '0'
    try:
        eng4, S4, A4, Rw4, loss4, g4, _ = _run(d)
    finally:
        del os.environ['PMBRL_REG_TREE']
    assert not eng4.info['reg'] and eng4.info['mm_mode'] == 1 and eng4.info['mm_parts'] == eng.info['mm_parts']
    print('   vs the latency-optimised family, same exchange: states %.2e grad %.2e' % (common.rel(S4, S), common.rel(g4, g)))
    assert common.rel(S4, S) < 5e-6 and common.rel(g4, g) < 6e-5 and common.rel(g4, g64.numpy()) < 1e-4
    # the same bits from run to run (fixed summation order at both levels)
    eng_b, S_b, A_b, Rw_b, loss_b, g_b, _ = _run(d)
    assert np.array_equal(S, S_b) and np.array_equal(g, g_b)


def _spanning_forms(d, groups, split):
    import os
    from oracle import ref_torch as R
    eng, S, A, Rw, loss, g, _ = _run(d)
    if split is not None:
        # (the split form now runs on the register-resident family: its adjoint of the moment matching is products with
        #  the stashed L^-1 on the fp64 matrix core where the barrier form solves triangular systems in scalar fp64 --
        #  2.6e-5 apart on the gradient, each inside 1e-4 of the fp64 oracle below / in test_full_size_matches_oracle)
        assert common.rel(split[0], S) < 5e-6 and common.rel(split[1], g) < 6e-5
    # every workgroup is resident at once here: ONE launch per sweep, a device-wide barrier per step
    assert eng.info['mm_mode'] == 3 and eng.info['rows_per_wg'] == 16 and eng.info['mm_grid'] == 1
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(8)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True,
                                            meta['mm_groups'], z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4
    os.environ['PMBRL_MM_MODE2'] = '1'
    try:
        eng2, S2, A2, Rw2, loss2, g2, _ = _run(d)
    finally:
        del os.environ['PMBRL_MM_MODE2']
    assert eng2.info['mm_mode'] == 2
    assert common.rel(S2, S) < 1e-6 and common.rel(g2, g) < 1e-5
    # the per-step-launch form of the same path (what runs when the workgroups outnumber the CUs):
    # same arithmetic in the same order, so the same bits
    os.environ['PMBRL_MM_PERSTEP'] = '1'
    try:
        eng4, S4, A4, Rw4, loss4, g4, _ = _run(d)
    finally:
        del os.environ['PMBRL_MM_PERSTEP']
    assert eng4.info['mm_mode'] == 3 and eng4.info['mm_grid'] == 0
    assert np.array_equal(S4, S) and np.array_equal(g4, g)
    # 32-row workgroups (two row tiles): same path, different tiling
    eng3, S3, A3, Rw3, loss3, g3, _ = _run(d, rows_per_wg_hint=32)
    assert eng3.info['mm_mode'] == 3 and eng3.info['rows_per_wg'] == 32
    assert common.rel(S3, S) < 1e-6 and common.rel(g3, g) < 1e-5


def test_wide_network_general_family_matches_oracle():
    """Hidden layers wider than the latency-optimised family takes (> 16 tiles) run the general
    kernels; with leading dimension >= 240 their K-split partial tiles live in the output buffer's
    free columns instead of a dedicated LDS region.  Against the fp64 oracle."""
    from oracle import ref_torch as R
    d = _problem('wide_small')
    eng, S, A, Rw, loss, g, _ = _run(d, lean=False)
    assert eng.info['fast'] == 0 and eng.info['LD'] >= 240
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, False, False, None,
                                            z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4


@pytest.mark.parametrize('config', ['wide_actions', 'mid_actions'])
def test_wide_action_vectors_match_oracle(config):
    """U = 12 and U = 6: the reward launch's instances with the action cost's quadratic forms unrolled to 16 / 8
    (pm_reward_all_kernel<16> / <8>; the cart-pole shapes run <4>, C5 <8>) -- trajectory, rewards, loss and gradient against
    the fp64 oracle at the plain bars."""
    from oracle import ref_torch as R
    d = _problem(config)
    eng, S, A, Rw, loss, g, _ = _run(d)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                            meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert common.rel(A, torch.stack(A64).detach().numpy()) < 2e-5
    assert common.rel(Rw, torch.stack(R64).detach().numpy().reshape(Rw.shape)) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4


def test_c5_shape_matches_oracle():
    """BASELINE.json configs[4] (D=32, U=8, 3 x 512 both nets, H=100, generic reward) on the general
    kernel family, at a row count the fp64 oracle finishes in seconds (8 particles x 64 samples)."""
    from oracle import ref_torch as R
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem('stress32', seed=0, data_seed=0, P=8, S=64))
    assert int(d['H']) == 100 and d['x0'].shape == (512, 32) and d['pol_W1'].shape == (512, 512)
    eng, S, A, Rw, loss, g, _ = _run(d)
    assert eng.info['fast'] == 0
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, False, False, None,
                                            z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert common.rel(A, torch.stack(A64).detach().numpy()) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4
    # 32-row workgroups (what the full-size configuration runs): same results up to summation order
    eng2, S2, A2, Rw2, loss2, g2, _ = _run(d, rows_per_wg_hint=32)
    assert eng2.info['rows_per_wg'] == 32
    assert common.rel(S2, S) < 1e-6 and common.rel(g2, g) < 1e-5


@pytest.mark.parametrize('parts', [4, 2, 1], ids=['register_resident_4_parts', 'split_groups', 'whole_groups'])
def test_c4_full_size_matches_oracle(parts):
    """BASELINE.json configs[3], one GPU's share: D=6, 100 x 50 rows in 50-row moment-matching groups,
    the real horizon H=60, against the fp64 oracle -- with every group in four parts of <= 13 rows on the
    register-resident family (round 5's default at this size: 400 workgroups in two launches of 50 whole groups each,
    csrc/pmbrl_reg_mm.h), split over two 32-row workgroups of the latency-optimised family (the default before;
    PMBRL_MM_NO_BATCH=1) and with one 64-row workgroup per group (what a plan with more than 128 groups uses)."""
    from oracle import ref_torch as R
    d = _problem('dcartpole_mm')
    assert int(d['H']) == 60 and d['x0'].shape == (5000, 6)
    if parts == 1:
        os.environ['PMBRL_MM_PARTS'] = '1'
    if parts == 2:
        os.environ['PMBRL_MM_NO_BATCH'] = '1'
    try:
        eng, S, A, Rw, loss, g, _ = _run(d)
    finally:
        os.environ.pop('PMBRL_MM_PARTS', None)
        os.environ.pop('PMBRL_MM_NO_BATCH', None)
    assert eng.info['fast'] == 1 and eng.info['mm_mode'] == 1 and eng.info['mm_parts'] == parts
    assert eng.info['rows_per_wg'] == (50 + parts - 1) // parts and eng.info['row_tiles'] == {4: 1, 2: 2, 1: 4}[parts]
    assert bool(eng.info['reg']) == (parts == 4) and eng.reg_calls() == ((1, 1) if parts == 4 else (0, 0))
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True,
                                            meta['mm_groups'], z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert common.rel(Rw.reshape(60, 5000), torch.stack(R64).detach().numpy().reshape(60, 5000)) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4


# ---------------------------------------------------------------------------
# BASELINE.json configs[4] (C5) at its per-GPU size: 256 particles x 64 samples = 16 384 rows, H = 100, with and
# without the moment matching BASELINE.md's table gives it (mm_groups = particles: 64-row groups, 32 x 32 covariances)
# ---------------------------------------------------------------------------
def _sub_rows(d, n):
    """The first n rows of problem d as a problem of its own (shared inputs stay whole)."""
    B = d['x0'].shape[0]
    e = dict(d)
    for k in list(e):
        v = np.asarray(e[k]) if not isinstance(e[k], (str, bool, int, float)) else None
        if v is not None and (k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and v.ndim == 2 and v.shape[0] == B)):
            e[k] = v[:n]
    if int(d['mm_groups']):
        e['mm_groups'] = np.asarray(int(d['mm_groups']) * n // B)
    return e


def _oracle_on_first_rows(d, n, threads=16):
    """fp64 oracle on the first n rows (whole moment-matching groups) of the global batch d: the cyclic noise of
    utils/rollout.py:53-59 indexed with the GLOBAL batch size, the loss a mean over the global batch."""
    from oracle import ref_torch as R
    B = d['x0'].shape[0]
    e = _sub_rows(d, n)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(e, torch.float64)
    torch.set_num_threads(threads)
    orig = R.get_z_rnd
    R.get_z_rnd = lambda z, i, m: z[torch.arange(i, i + m) % B]
    try:
        l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                                meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
    finally:
        R.get_z_rnd = orig
    return torch.stack(S64).detach().numpy(), float(l64) * n / B, g64.numpy() * (n / B)


def test_c5_mm_shape_matches_oracle():
    """C5 WITH moment matching (D = 32: a 32 x 32 covariance per 64-row group and step) at 8 particles x 64 samples,
    the real horizon H = 100, against the fp64 oracle (utils/rollout.py:20-29 over the 3 x 512 networks)."""
    from oracle import ref_torch as R
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem('stress32_mm', seed=0, data_seed=0, P=8, S=64))
    assert int(d['H']) == 100 and d['x0'].shape == (512, 32) and int(d['mm_groups']) == 8
    eng, S, A, Rw, loss, g, _ = _run(d)
    assert eng.info['fast'] == 0 and eng.info['mm_mode'] in (1, 2)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True, 8, z_mm, z_rr)
    e_s, e_g = common.rel(S, torch.stack(S64).detach().numpy()), common.rel(g, g64.numpy())
    print('C5 mm 512 rows: states %.2e loss %.2e grad %.2e' % (e_s, abs(loss - float(l64)) / abs(float(l64)), e_g))
    assert e_s < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert e_g < 1e-4


def _c5_full(config):
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem(config, seed=0, data_seed=0))
    assert int(d['H']) == 100 and d['x0'].shape == (16384, 32)
    return d


def _masked(gw, n):
    m = torch.zeros_like(gw)
    m[:, :n] = gw[:, :n]
    return m


def _fp32_floor_on_first_rows(d, n, g64):
    """The reference's own arithmetic (the oracle in fp32) against its fp64 run on the same rows: the noise floor of
    any fp32-class implementation of this gradient."""
    from oracle import ref_torch as R
    B = d['x0'].shape[0]
    e = _sub_rows(d, n)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(e, torch.float32)
    orig = R.get_z_rnd
    R.get_z_rnd = lambda z, i, m: z[torch.arange(i, i + m) % B]
    try:
        _, g32, _ = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'], meta['mm_rewards'],
                                meta['mm_groups'], z_mm, z_rr)
    finally:
        R.get_z_rnd = orig
    return common.rel(g32.numpy() * (n / B), g64)


def test_c5_full_size_parity_and_properties():
    """C5 at its per-GPU size (16 384 rows x H = 100, no moment matching: the configuration bench.py --config
    stress32 times).  Rows are independent, so (a) the first 512 rows' trajectories and the gradient of the loss
    restricted to them are the fp64 oracle's on those rows; (b) reversing the row order reverses the trajectories
    bit for bit; (c) 16- and 32-row workgroups agree; (d) the gradient is linear in the loss weights.

    The gradient bound of (a).  Trajectories agree to 3e-7.  The policy gradient of a ReLU network is not continuous
    in the pre-activations: a unit whose pre-activation lies within rounding of zero is active in one arithmetic and
    inactive in another, and each such unit moves the gradient by a finite amount.  Over 512 rows x 100 steps x 3072
    units a handful of them exist in ANY fp32-class arithmetic -- the reference's own fp32 run differs from its fp64
    run by 9.6e-5 on these rows (measured below, not assumed).  So the bar is the north star's 1e-4 on top of the
    reference's own fp32-vs-fp64 distance, for the exact-fp32 path AND for the default split arithmetic -- the one
    bench.py --config stress32 times.  (Round 3's split arithmetic sat at 2.3e-4 here: the weights' low fp16 piece was
    subnormal for every |w| < 0.125 and kept 8 of its 11 bits at this width; since round 4 it is stored scaled by 2^11
    -- pmbrl_split.h, PM_F16_LO_SCALE; tools/c5_precision_study.py -- and the two-piece forward is fp32-class.)"""
    d = _c5_full('stress32')
    eng, S, A, Rw, loss, g, gw = _run(d)
    # (round 4: 64-row workgroups on the in-place 512-wide layers of pmbrl_wide.h -- one workgroup per CU)
    assert eng.info['fast'] == 0 and eng.info['rows_per_wg'] == 64 and eng.info['inplace'] == 2
    S64, l64, g64 = _oracle_on_first_rows(d, 512)
    floor32 = _fp32_floor_on_first_rows(d, 512, g64)
    g_sub = eng.backward(_masked(gw, 512))[0].cpu().numpy().copy()
    e_s, e_g = common.rel(S[:, :512], S64), common.rel(g_sub, g64)
    # the exact-fp32 path on the same 512 rows (a shard of the same global batch)
    from prob_mbrl_amd import problem as PB
    e32, a32, _ = PB.engine_from_problem(_sub_rows(d, 512), DEV, B_global=16384, row_offset=0, precision='f32')
    e32.forward(**a32)
    g_f32 = e32.backward(gw[:, :512].contiguous())[0].cpu().numpy()
    e_g32 = common.rel(g_f32, g64)
    del e32
    print('C5 16384 rows, first 512 vs oracle: states %.2e grad %.2e (exact-fp32 path %.2e, reference fp32 vs fp64 %.2e)'
          % (e_s, e_g, e_g32, floor32))
    assert e_s < 2e-5
    # the benchmarked (default, split) arithmetic holds the PLAIN bar (measured 4.3e-5): a regression towards round 3's
    # 2.3e-4 must not pass.  The widened bar stays only for the exact-fp32 comparison path, whose distance to fp64 is the
    # reference's own kind of rounding (5.6e-5 measured, against the reference's 9.6e-5)
    assert e_g < 1e-4, e_g
    assert e_g32 < 1e-4 + floor32
    # (d) linearity: the rest of the rows' gradient adds up to the whole
    g_rest = eng.backward(gw - _masked(gw, 512))[0].cpu().numpy().copy()
    assert common.rel(g_sub + g_rest, g) < 2e-6
    # (c) 16- and 32-row workgroups (the two-buffer form).  Its hidden layers are the wide layers' bit for bit, its
    #     heads sum their K in another order (K-split over the waves): trajectories agree to fp32 rounding, and between
    #     two such arithmetics the gradient at this size carries the same handful of borderline ReLU units as (a) --
    #     the bar is (a)'s
    for hint in (16, 32):
        eng16, S16, _, _, _, g16, _ = _run(d, rows_per_wg_hint=hint)
        assert eng16.info['rows_per_wg'] == hint and eng16.info['inplace'] == 0
        print('C5 %d-row two-buffer form vs the wide layers: states %.2e grad %.2e' % (hint, common.rel(S16, S), common.rel(g16, g)))
        assert common.rel(S16, S) < 2e-6 and common.rel(g16, g) < 1e-4 + floor32
        del eng16
    # (b) row reversal
    B = d['x0'].shape[0]
    e = dict(d)
    for k in list(e):
        v = np.asarray(e[k]) if not isinstance(e[k], (str, bool, int, float)) else None
        if v is not None and (k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and v.ndim == 2 and v.shape[0] == B)):
            e[k] = v[::-1].copy()
    _, Sr, Ar, _, _, gr, _ = _run(e)
    assert np.array_equal(S[:, ::-1], Sr) and np.array_equal(A[:, ::-1], Ar)
    assert common.rel(gr, g) < 2e-6


def test_c5_mm_full_size_parity_and_properties():
    """C5 with mm_groups = 256 at its per-GPU size (16 384 rows x H = 100: bench.py --config stress32_mm).  Groups
    are independent, so (a) the first 8 groups' trajectories and the gradient of the loss restricted to them are the
    fp64 oracle's on those 512 rows (noise rows taken with the global row index, as the kernels and the reference
    do); (b) the same 8 groups run as a shard of the global batch give the same bits; (c) the gradient is linear
    in the loss weights."""
    from prob_mbrl_amd import problem as PB
    d = _c5_full('stress32_mm')
    eng, S, A, Rw, loss, g, gw = _run(d)
    assert eng.info['fast'] == 0 and int(d['mm_groups']) == 256
    S64, l64, g64 = _oracle_on_first_rows(d, 512)
    g_sub = eng.backward(_masked(gw, 512))[0].cpu().numpy().copy()
    e_s, e_g = common.rel(S[:, :512], S64), common.rel(g_sub, g64)
    print('C5 mm 16384 rows, first 8 groups vs oracle: states %.2e grad %.2e' % (e_s, e_g))
    assert e_s < 2e-5 and e_g < 1e-4
    g_rest = eng.backward(gw - _masked(gw, 512))[0].cpu().numpy().copy()
    assert common.rel(g_sub + g_rest, g) < 2e-6
    # (b) the first 8 groups as a shard of the 16 384-row global batch
    # (in the same kernel form as the whole batch -- 64-row workgroups on the wide layers: a 512-row plan would take
    #  the two-buffer form by itself, whose policy head sums its K in another order)
    sh, args, _ = PB.engine_from_problem(_sub_rows(d, 512), DEV, B_global=16384, row_offset=0,
                                         rows_per_wg_hint=eng.info['rows_per_wg'])
    assert sh.info['inplace'] == eng.info['inplace']
    Ss, As_, Rs = sh.forward(**args)
    gs = sh.backward(gw[:, :512].contiguous())[0].cpu().numpy()
    assert sh.valid_steps() == 100
    assert np.array_equal(Ss.cpu().numpy(), S[:, :512]) and np.array_equal(As_.cpu().numpy(), A[:, :512])
    assert common.rel(gs, g_sub) < 2e-6


def test_device_detects_a_mid_horizon_failure_and_continues_on_the_truncated_horizon():
    """A failure the DEVICE finds (no status word written by the test): one state dimension, decoupled from the
    networks and the reward, drifts by 5e37 per step and overflows fp32 while step 6 computes x_7; the state moment
    matching of step 6 sees a non-finite covariance -- the condition under which the reference's Cholesky raises
    (utils/rollout.py:116-127) -- and the caller keeps the 6 completed steps (:154-157; more than 5).  The kept
    trajectory, the loss and the gradient are the fp64 oracle's run to 6 steps.  (WHICH step overflows depends on
    the arithmetic: the reference's own fp32 run would overflow the SUM of a group's 25 values of 5e37 in its mean
    two steps in; the device takes its moments in fp64, so its first non-finite value is x_7 itself.  What is pinned
    here is the device's detection and the continuation after it, against the oracle in fp64.)"""
    from oracle import ref_torch as R
    d = dict(common.load('mmg_h40'))
    H, B = int(d['H']), d['x0'].shape[0]
    k = 3
    for key in ('pol_W0', 'dyn_W0'):
        W = np.asarray(d[key]).copy()
        W[:, k] = 0.0
        d[key] = W
    C = np.asarray(d['rew_C']).copy()
    assert bool(d['rew_expand'])
    # columns of the expanded state [others (0, 1, 3) | sin | cos]: the decoupled dimension 3 is column 2
    C[:, 2] = 0.0
    d['rew_C'] = C
    my = np.asarray(d['dyn_my'], dtype=np.float32).copy()
    my[k] = 5e37
    d['dyn_my'] = my
    # the general family on fp16 pieces cannot hold a normalised input of 1e37: it says so at the first step that sees
    # one (rollout() / mc_pilco then re-run in fp32, tests/test_gpu_api.py) instead of rolling on with a NaN that
    # vanishes in the ReLU
    eng, args, _ = common.engine_from_fixture(d, DEV, force_generic=True, precision='split_f16')
    eng.forward(**args)
    assert eng.valid_steps() <= 1
    for generic, prec in ((False, None), (True, 'f32')):
        eng, args, _ = common.engine_from_fixture(d, DEV, force_generic=generic, precision=prec)
        S, A, Rw = eng.forward(**args)
        n = eng.valid_steps()
        assert n == 6, n                                # found by the device, at step 6 of 40
        gw = torch.tensor(common.loss_weights(d, B), device=DEV)
        loss = float((Rw[:n].reshape(n, B) * gw[:n]).sum())
        g, _, _ = eng.backward(gw)
        g = g.cpu().numpy()
        assert np.all(np.isfinite(g))
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
        l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, H, gamma, True, True, True, meta['mm_groups'],
                                                z_mm, z_rr, n_steps=n)
        S64 = torch.stack(S64).detach().numpy()
        keep = [j for j in range(S64.shape[-1]) if j != k]
        Sd = S[:n + 1].cpu().numpy()
        assert common.rel(Sd[..., keep], S64[..., keep]) < 2e-5
        assert common.rel(Sd[..., k], S64[..., k]) < 1e-6
        assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
        assert common.rel(g, g64.numpy()) < 1e-4


@pytest.mark.parametrize('config', ['cartpole_nomm', 'cartpole_mm'])
def test_fused_tail_walks_the_separate_launches_trajectory_at_full_size(config):
    """The pair of tests/test_gpu_kernels.py::test_fused_tail_matches_the_separate_launches at a size where a trajectory
    comparison means something: three iterations through the fused tail (pmbrl_plan_set_loss + pmbrl_rollout_bwd_adam:
    loss, adjoint, dW, clip, device-guarded Adam) and three through the separate calls (weighted_sum, backward, clip_adam),
    EACH ON ITS OWN trajectory from the same start.  At 2 500 rows x 40 steps a hidden unit changing sides under a
    1e-7 parameter difference is one of 10^7 row-step-units, not one of 37 rows: losses agree to 1e-6, the parameters after
    three steps to 2e-3 of a step (lr).  (Measured: identical bits on both configurations -- the device-side step forms its
    bias corrections in double like the host-side one.)"""
    from prob_mbrl_amd import engine as E
    from prob_mbrl_amd import problem as PB
    d = _problem(config)
    B = d['x0'].shape[0]
    gw = torch.tensor(PB.loss_weights(d, B), device=DEV)
    lr = 1e-3
    out = []
    for fused in (True, False):
        eng, args, _ = PB.engine_from_problem(d, DEV)
        p = args['pol_flat'].clone()
        args['pol_flat'] = p
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        step = torch.zeros(1, dtype=torch.int64, device=DEV)
        losses = []
        if fused:
            loss_buf = eng.set_loss(gw)
            for it in range(1, 4):
                eng.forward(**args)
                eng.backward(gw, adam=dict(params=p, exp_avg=m, exp_avg_sq=v, step=step, lr=lr, betas=(0.9, 0.999),
                                           eps=1e-8, max_norm=1.0))
                losses.append(float(loss_buf))
            assert int(step.item()) == 3
        else:
            for it in range(1, 4):
                _, _, R = eng.forward(**args)
                losses.append(float(eng.weighted_sum(R, gw)))
                g = eng.backward(gw)[0].clone()
                E.clip_adam(p, g, m, v, it, lr, max_norm=1.0)
        assert eng.valid_steps() == int(d['H']) and eng.reg_calls()[0] >= 3
        out.append((np.array(losses), p.cpu().numpy().copy(), m.cpu().numpy().copy()))
    (lf, pf, mf), (ls, ps, ms) = out
    print('%s: losses %s vs %s; max |dp| / lr %.2e; moments rel %.2e' %
          (config, lf, ls, np.abs(pf - ps).max() / lr, common.rel(mf, ms)))
    assert np.allclose(lf, ls, rtol=1e-6)
    assert np.abs(pf - ps).max() <= 2e-3 * lr
    assert common.rel(mf, ms) < 1e-4
