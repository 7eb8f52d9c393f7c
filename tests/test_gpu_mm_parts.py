"""In-kernel moment matching with a group SPLIT over several workgroups (pmbrl.hip: mm_parts).  Instances of
compile-time state width: every part sums the fp64 Gram tile over its own rows and the parts exchange the sums as
data-tagged granules (pmbrl_fast.h: pm_xch_put / pm_xch_get).  Generic instances (and PMBRL_MM_XCH=0): the
workgroups of a group exchange their rows through HBM, meet at a group-local flag barrier (pm_group_sync) and each
factors the whole group.  On by default where a group needs a 64-row workgroup (the double cart-pole
shape: two 25-row workgroups instead); forced here on the ordinary fixtures (PMBRL_MM_PARTS=n) and compared with
whole groups per workgroup and with the fp64 reference numbers."""
import contextlib
import os

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_TRAJ, TOL_GRAD = 2e-5, 1e-4


@contextlib.contextmanager
def parts(n):
    old = os.environ.get('PMBRL_MM_PARTS')
    os.environ['PMBRL_MM_PARTS'] = str(n)
    try:
        yield
    finally:
        if old is None:
            del os.environ['PMBRL_MM_PARTS']
        else:
            os.environ['PMBRL_MM_PARTS'] = old


def run(d, n, precision=None, n_valid=None, no_shaped=False):
    dev = torch.device(DEV)
    with parts(n):
        eng, args, _ = common.engine_from_fixture(d, dev, precision=precision, no_shaped=no_shaped)
    S, A, Rw = eng.forward(**args)
    nv = eng.valid_steps()
    if n_valid is not None:
        eng.status[0] = n_valid
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    g, gx0, agn = eng.backward(gw, want_x0=True, want_agn=True)
    torch.cuda.synchronize()
    return eng, nv, S.cpu().numpy(), Rw.cpu().numpy(), g.cpu().numpy().copy(), gx0.cpu().numpy().copy()


@pytest.mark.parametrize('precision', ['f32', 'split_f16'])
@pytest.mark.parametrize('name,n', [('mmg_d4', 2), ('mmg_d4', 3), ('mmg_d4', 4), ('dcp_d6_mmg', 2), ('dcp_d6_mmg', 3),
                                    ('dcp_d6_mmg', 4), ('full200_mmg', 2), ('full200_mmg', 3), ('full200_mmg', 5),
                                    ('mmg_h40', 2), ('mm1_b100_h40', 4), ('mm1_b100', 7), ('mmg_m80', 3), ('mmg_m80', 5),
                                    ('angles_dcp_mmg', 2)])
def test_split_groups_match_whole_groups(name, n, precision):
    d = common.load(name)
    e1, nv1, S1, R1, g1, x1 = run(d, 1, precision)
    if not e1.info['fast']:
        pytest.skip('general kernel family')
    e2, nv2, S2, R2, g2, x2 = run(d, n, precision)
    G = max(1, int(d['mm_groups']))
    M = d['x0'].shape[0] // G
    assert M <= 128      # (beyond 64 rows the one-wave routines walk the rows in strides)
    assert e1.info['mm_parts'] == 1
    # (unequal parts: the last workgroup of a group takes what is left of its M rows)
    assert e2.info['mm_parts'] == n and e2.info['mm_mode'] == 1 and e2.info['n_wg'] == n * G, e2.info
    assert e2.info['rows_per_wg'] == -(-M // n)
    assert nv1 == nv2 == int(d['H'])
    # the same group statistics from the same rows: only the summation order inside the GEMMs' K-split differs
    assert common.rel(S2, S1) < 5e-6 and common.rel(g2, g1) < 2e-5
    assert common.rel(S2, d['ref64_states']) < TOL_TRAJ
    assert common.rel(R2.reshape(d['ref64_rewards'].shape), d['ref64_rewards']) < TOL_TRAJ
    assert common.rel(g2, d['ref64_grad']) < TOL_GRAD
    assert common.rel(x2, x1) < 2e-5


def test_split_groups_truncated_horizon():
    """trunc_mm: one group of 30 rows, H = 12, valid horizon 8: two workgroups of 15 rows."""
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    e1, _, S1, _, g1, x1 = run(d, 1, n_valid=n)
    e2, _, S2, _, g2, x2 = run(d, 2, n_valid=n)
    assert e2.info['mm_parts'] == 2 and e2.info['rows_per_wg'] == 15
    assert common.rel(g2, g1) < 2e-5 and common.rel(g2, d['ref64_grad']) < TOL_GRAD
    assert common.rel(S2[:n + 1], d['ref64_states']) < TOL_TRAJ


def test_number_of_parts_chosen_by_the_plan():
    """The fewest parts that bring a part down to 16 rows with every workgroup resident, else to 32 rows."""
    assert 'PMBRL_MM_PARTS' not in os.environ
    dev = torch.device(DEV)
    for name, parts, rows in (('mm1_b100', 7, 15), ('mmg_m80', 5, 16), ('mmg_h40', 2, 13), ('dcp_d6_mmg', 1, 12),
                              ('trunc_mm', 2, 15)):
        d = common.load(name)
        eng = common.engine_from_fixture(d, dev)[0]
        assert (eng.info['mm_parts'], eng.info['rows_per_wg']) == (parts, rows), (name, eng.info)


def test_double_cartpole_shape_splits_its_groups_by_default():
    from prob_mbrl_amd import problem as PB
    assert 'PMBRL_MM_PARTS' not in os.environ
    pr = PB.synthetic_problem('dcartpole_mm', seed=0, data_seed=0)
    eng = PB.engine_from_problem(pr, torch.device(DEV))[0]
    # (round 5: four parts of <= 13 rows on the register-resident family, 400 workgroups launched as two batches of 50
    #  whole groups -- the statistics exchange needs ONE group's workgroups resident together, not all of them)
    assert eng.info['mm_parts'] == 4 and eng.info['rows_per_wg'] == 13 and eng.info['n_wg'] == 400 and eng.info['reg'], eng.info
    os.environ['PMBRL_MM_NO_BATCH'] = '1'
    try:
        eng = PB.engine_from_problem(pr, torch.device(DEV))[0]
    finally:
        del os.environ['PMBRL_MM_NO_BATCH']
    assert eng.info['mm_parts'] == 2 and eng.info['rows_per_wg'] == 25 and eng.info['n_wg'] == 200 and not eng.info['reg'], eng.info
    # 25-row groups: 13 + 12 rows in two 16-row workgroups instead of one 32-row workgroup
    pr = PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0)
    eng = PB.engine_from_problem(pr, torch.device(DEV))[0]
    assert eng.info['mm_parts'] == 2 and eng.info['rows_per_wg'] == 13 and eng.info['n_wg'] == 200, eng.info
    # more groups than half the CUs: still two parts on the register-resident family, in two launches (before round 5, and
    # for shapes that family does not take: whole groups per 32-row workgroup)
    pr = PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0, P=160)
    eng = PB.engine_from_problem(pr, torch.device(DEV))[0]
    assert eng.info['mm_parts'] == 2 and eng.info['rows_per_wg'] == 13 and eng.info['n_wg'] == 320 and eng.info['reg'], eng.info
    os.environ['PMBRL_MM_NO_BATCH'] = '1'
    try:
        eng = PB.engine_from_problem(pr, torch.device(DEV))[0]
    finally:
        del os.environ['PMBRL_MM_NO_BATCH']
    assert eng.info['mm_parts'] == 1 and eng.info['rows_per_wg'] == 25, eng.info


def test_split_groups_replay_in_a_graph():
    """The flag barriers count from zero in every launch (their flags are cleared by a memset node of the graph)."""
    d = common.load('dcp_d6_mmg')

    def go(use_graph):
        with parts(2):
            eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        assert eng.info['mm_parts'] == 2
        gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=DEV)
        out = []

        def step():
            eng.forward(**args)
            return eng.backward(gw)[0]

        if not use_graph:
            for _ in range(3):
                g = step()
            return g.cpu().numpy().copy()
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            step()
        torch.cuda.current_stream().wait_stream(s)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            g = step()
        for _ in range(2):
            graph.replay()
        torch.cuda.synchronize()
        return g.cpu().numpy().copy()

    assert np.array_equal(go(False), go(True))


def test_forward_only_graph_replays_do_not_meet_their_own_tags():
    """A recorded forward call of the register-resident family is replayed with the launch generation it was recorded
    with.  Replayed twice with no adjoint call between, the second replay would find the first one's granules (same tags)
    and could accept sums a partner has not re-published yet -- so a recording carries the fill of the exchange buffer.
    Forward-only replays with the policy changed between them must give what eager calls give, bit for bit."""
    d = common.load('full200_mmg')
    eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
    assert eng.info['reg'] and eng.info['mm_parts'] >= 2, eng.info
    p0 = args['pol_flat'].clone()
    scales = [1.0, 0.97, 1.02, 0.99, 1.0, 1.01]

    def eager(sc):
        args['pol_flat'].copy_(p0 * sc)
        S, A, R = eng.forward(**args)
        torch.cuda.synchronize()
        return S.cpu().numpy().copy(), R.cpu().numpy().copy()

    want = [eager(sc) for sc in scales]
    assert not np.array_equal(want[0][0], want[1][0])
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(**args)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        S, A, R = eng.forward(**args)
    for rep in range(3):
        for sc, (S_w, R_w) in zip(scales, want):
            args['pol_flat'].copy_(p0 * sc)
            graph.replay()
            torch.cuda.synchronize()
            assert np.array_equal(S.cpu().numpy(), S_w) and np.array_equal(R.cpu().numpy(), R_w), (rep, sc)
    # ... and eager calls after the replays (a new generation over the recording's leftover tags)
    S2, R2 = eager(scales[1])
    assert np.array_equal(S2, want[1][0]) and np.array_equal(R2, want[1][1])


@pytest.mark.parametrize('name,n', [('mmg_h40', 2), ('mm1_b100_h40', 4), ('full200_mmg', 3)])
def test_statistics_exchange_matches_row_exchange(name, n):
    """The two forms of a split group on the shapes that have a compile-time-width instance: sums exchanged as
    granules (default) against rows + flag barrier (PMBRL_MM_XCH=0, generic instance)."""
    d = common.load(name)
    e1, nv1, S1, R1, g1, x1 = run(d, n)
    assert 'PMBRL_MM_XCH' not in os.environ
    os.environ['PMBRL_MM_XCH'] = '0'
    try:
        e2, nv2, S2, R2, g2, x2 = run(d, n)
    finally:
        del os.environ['PMBRL_MM_XCH']
    assert e1.info['mm_parts'] == e2.info['mm_parts'] == n and nv1 == nv2 == int(d['H'])
    assert common.rel(S2, S1) < 5e-6 and common.rel(g2, g1) < 2e-5 and common.rel(x2, x1) < 2e-5
    assert common.rel(g1, d['ref64_grad']) < TOL_GRAD and common.rel(g2, d['ref64_grad']) < TOL_GRAD


@pytest.mark.parametrize('name', ['full200_mmg', 'dcp200_mmg50'])
def test_noise_table_in_the_pack_launch_matches_its_own_launch(name, monkeypatch):
    """Round 6: on the register-resident family the noise table of split groups (mean and 1 / std of every group's noise
    rows per step) is formed by extra workgroups of the weight-pack launch (pm_reg_pack_kernel<true>, pm_ztab_block);
    PMBRL_ZTAB_MERGE=0 forms it in the launch of its own it had.  One definition of the arithmetic: identical results."""
    d = common.load(name)
    dev = torch.device(DEV)

    def once():
        eng, args, _ = common.engine_from_fixture(d, dev)
        S, A, Rw = eng.forward(**args)
        gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
        g, _, _ = eng.backward(gw)
        torch.cuda.synchronize()
        assert eng.info['reg'] and eng.reg_calls() == (1, 1) and eng.info['mm_parts'] > 1, eng.info
        return S.cpu().numpy(), Rw.cpu().numpy(), g.cpu().numpy().copy()

    S1, R1, g1 = once()
    monkeypatch.setenv('PMBRL_ZTAB_MERGE', '0')
    S0, R0, g0 = once()
    assert np.array_equal(S1, S0) and np.array_equal(R1, R0) and np.array_equal(g1, g0)

