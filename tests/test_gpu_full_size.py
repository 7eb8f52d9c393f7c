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
    for kw, lean in ((dict(no_shaped=True), True), (dict(), False), (dict(no_shaped=True), False)):
        out = _run(d, lean=lean, **kw, **pk)
        assert np.array_equal(ref[1], out[1]) and np.array_equal(ref[2], out[2]) and np.array_equal(ref[3], out[3])
        assert np.array_equal(ref[5], out[5])


def test_gradient_is_linear_in_loss_weights():
    d = _problem('cartpole_nomm')
    eng, S, A, Rw, loss, g, gw = _run(d)
    g2, _, _ = eng.backward(2.0 * gw)
    assert common.rel(g2.cpu().numpy(), 2.0 * g) < 1e-6        # (not bit-exact: denormals are flushed)
    w1 = gw * torch.rand_like(gw)
    ga = eng.backward(w1)[0].cpu().numpy().copy()
    gb = eng.backward(gw - w1)[0].cpu().numpy().copy()
    assert common.rel(ga + gb, g) < 2e-6


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


def _spanning_forms(d, groups, split):
    import os
    from oracle import ref_torch as R
    eng, S, A, Rw, loss, g, _ = _run(d)
    if split is not None:
        assert common.rel(split[0], S) < 5e-6 and common.rel(split[1], g) < 2e-5
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


@pytest.mark.parametrize('parts', [2, 1], ids=['split_groups', 'whole_groups'])
def test_c4_full_size_matches_oracle(parts):
    """BASELINE.json configs[3], one GPU's share: D=6, 100 x 50 rows in 50-row moment-matching groups,
    the real horizon H=60, against the fp64 oracle -- with every group split over two 32-row workgroups (the
    default at this size) and with one 64-row workgroup per group (what a plan with more than 128 groups uses)."""
    from oracle import ref_torch as R
    d = _problem('dcartpole_mm')
    assert int(d['H']) == 60 and d['x0'].shape == (5000, 6)
    if parts == 1:
        os.environ['PMBRL_MM_PARTS'] = '1'
    try:
        eng, S, A, Rw, loss, g, _ = _run(d)
    finally:
        os.environ.pop('PMBRL_MM_PARTS', None)
    assert eng.info['fast'] == 1 and eng.info['mm_mode'] == 1 and eng.info['mm_parts'] == parts
    assert eng.info['rows_per_wg'] == 50 // parts and eng.info['row_tiles'] == (2 if parts == 2 else 4)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, True, True,
                                            meta['mm_groups'], z_mm, z_rr)
    assert common.rel(S, torch.stack(S64).detach().numpy()) < 2e-5
    assert common.rel(Rw.reshape(60, 5000), torch.stack(R64).detach().numpy().reshape(60, 5000)) < 2e-5
    assert abs(loss - float(l64)) <= 2e-5 * abs(float(l64))
    assert common.rel(g, g64.numpy()) < 1e-4
