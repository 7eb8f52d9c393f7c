"""dW GEMM behind the adjoint sweep (pmbrl.hip, pipe_K): the sweep cut into launches over descending step
ranges, the GEMM of each range on a second stream next to the sweep of the next one, partial rows written by
the first launch that owns a valid step and added to by the later ones.  The plan turns this on by itself only
for long sweeps with idle CUs (the double cart-pole shape); here it is forced with explicit ranges
(PMBRL_DW_PIPE=n0,n1,...) on the ordinary fixtures and compared with the one-launch form and the oracle's
numbers -- full and truncated horizons, both kernel families, eager and inside a hipGraph."""
import contextlib
import os

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'
TOL_GRAD = 1e-4


@contextlib.contextmanager
def pipe(spec):
    # (whole moment-matching groups per workgroup: groups split over workgroups meet at flag barriers and do not
    #  share the chip with a second kernel -- the plan leaves the pipeline off for them)
    new = {'PMBRL_DW_PIPE': spec, 'PMBRL_MM_PARTS': '1'}
    old = {k: os.environ.get(k) for k in new}
    os.environ.update(new)
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                del os.environ[k]
            else:
                os.environ[k] = v


def ranges(H, K):
    """K step counts summing to H, longest first."""
    w = np.arange(K, 0, -1, dtype=np.float64)
    n = np.maximum(1, np.floor(H * w / w.sum())).astype(int)
    n[0] += H - n.sum()
    assert n.sum() == H and (n > 0).all()
    return ','.join(str(int(v)) for v in n)


def run(d, spec, generic=False, n_valid=None, precision=None):
    dev = torch.device(DEV)
    with pipe(spec):
        eng, args, _ = common.engine_from_fixture(d, dev, force_generic=generic, precision=precision)
    S, A, Rw = eng.forward(**args)
    if n_valid is not None:
        eng.status[0] = n_valid
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    if n_valid is not None:
        gw[n_valid:] = float('nan')       # nothing may read the weights of the steps that do not count
    g, gx0, agn = eng.backward(gw, want_x0=True, want_agn=True)
    torch.cuda.synchronize()
    return eng, g.cpu().numpy().copy(), gx0.cpu().numpy().copy(), agn.cpu().numpy().copy()


@pytest.mark.parametrize('generic', [False, True], ids=['fast', 'generic'])
@pytest.mark.parametrize('K', [2, 3, 5])
@pytest.mark.parametrize('name', ['nomm_d4', 'mmg_d4', 'full200_nomm', 'dcp_d6_mmg', 'nomm_h40_disc'])
def test_pipelined_dw_matches_one_launch(name, K, generic):
    d = common.load(name)
    H = int(d['H'])
    if H < K:
        pytest.skip('horizon shorter than the number of ranges')
    e0, g0, x0, a0 = run(d, 'off', generic)
    e1, g1, x1, a1 = run(d, ranges(H, K), generic)
    assert e0.info['dw_pipe'] == 1
    if e1.info['mm_mode'] not in (0, 1):
        pytest.skip('per-step launch mode: not pipelined')
    # (split-operand GEMM with an odd number of chunks per step: range bounds move to even steps, ranges may merge)
    assert 2 <= e1.info['dw_pipe'] <= K, e1.info
    # the same sweeps (only the launch boundaries differ), the same products, another order of additions in dW
    assert common.rel(g1, g0) < 2e-6
    assert common.rel(x1, x0) < 1e-6 and common.rel(a1, a0) < 1e-6
    assert common.rel(g1, d['ref64_grad']) < TOL_GRAD


@pytest.mark.parametrize('generic', [False, True], ids=['fast', 'generic'])
@pytest.mark.parametrize('spec', ['5,4,3', '3,5,4', '2,2,8', '11,1', '1,1,1,1,1,1,1,5'])
def test_pipelined_dw_truncated_horizon(spec, generic):
    """trunc_mm: H = 12, the reference's rollout failed in step 8.  The ranges put the end of the valid horizon
    inside a range, on a boundary, and behind whole ranges (whose launches then have nothing to do, and the
    first launch with a valid step must WRITE its partial rows, not add to what an earlier iteration left)."""
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    e0, g0, x0, a0 = run(d, 'off', generic, n_valid=n)
    # a complete iteration first, so that every partial row holds stale sums
    e1, g_full, _, _ = run(d, spec, generic)
    if e1.info['mm_mode'] not in (0, 1):
        pytest.skip('per-step launch mode: not pipelined')
    assert 2 <= e1.info['dw_pipe'] <= spec.count(',') + 1
    e1, g1, x1, a1 = run(d, spec, generic, n_valid=n)
    assert np.all(np.isfinite(g1))
    assert common.rel(g1, g0) < 2e-6 and common.rel(g1, d['ref64_grad']) < TOL_GRAD
    assert common.rel(x1, x0) < 1e-6 and common.rel(a1[:n], a0[:n]) < 1e-6
    assert common.rel(g_full, d['ref64_grad']) > 1e-2


def test_pipelined_dw_reuses_rows_across_horizons():
    """One plan, one workspace: a complete iteration, then a truncated one, then a complete one again -- the
    write / add decision of every launch follows the status word of THAT iteration."""
    d = common.load('trunc_mm')
    n = int(d['fail_step'])
    dev = torch.device(DEV)
    with pipe('4,4,4'):
        eng, args, _ = common.engine_from_fixture(d, dev)
    assert eng.info['dw_pipe'] == 3
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    out = []
    for nv in (None, n, 3, 0, None):
        eng.forward(**args)
        if nv is not None:
            eng.status[0] = nv
        out.append(eng.backward(gw)[0].cpu().numpy().copy())
    assert np.array_equal(out[0], out[4])
    assert common.rel(out[1], d['ref64_grad']) < TOL_GRAD
    assert np.all(np.isfinite(out[2])) and common.rel(out[2], out[1]) > 1e-3
    assert np.all(out[3] == 0.0)


def test_pipelined_dw_in_a_graph():
    """The second stream joins the capture through its event wait and leaves it through the join before the
    last GEMM launch: the iteration still records into one hipGraph and replays bit-identically."""
    from prob_mbrl_amd import engine as E
    d = common.load('full200_nomm')
    H = int(d['H'])

    def go(n, use_graph):
        with pipe(ranges(H, 3)):
            eng, args, _ = common.engine_from_fixture(d, torch.device(DEV))
        assert eng.info['dw_pipe'] == 3
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
                step()
            torch.cuda.current_stream().wait_stream(s)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                step()
            for _ in range(n - 1):
                graph.replay()
        torch.cuda.synchronize()
        return params.cpu().numpy().copy()

    assert np.array_equal(go(5, False), go(5, True))


def test_long_sweeps_are_pipelined_by_default():
    """64-row workgroups with idle CUs (the double cart-pole shape) get the pipeline without being asked; the
    16-row cart-pole shape does not (a launch boundary costs more than the overlap returns there)."""
    from prob_mbrl_amd import problem as PB
    dev = torch.device(DEV)
    assert 'PMBRL_DW_PIPE' not in os.environ
    for name, want in (('dcartpole_mm', 4), ('cartpole_nomm', 1)):
        pr = PB.synthetic_problem(name, seed=0, data_seed=0)
        os.environ['PMBRL_MM_PARTS'] = '1'       # (64-row workgroups: one per 50-row group)
        try:
            eng = PB.engine_from_problem(pr, dev)[0]
        finally:
            del os.environ['PMBRL_MM_PARTS']
        assert eng.info['dw_pipe'] == want, (name, eng.info)


@pytest.mark.parametrize('spec', ['off', '9,7,4'])
@pytest.mark.parametrize('n_valid', [None, 13, 7, 0])
def test_wide_layer_gemm_matches_block_gemm(spec, n_valid):
    """The 512 x 512 layers of the stress shape go through pm_dw_wide_kernel (LDS-staged 128 x 128 tiles); the
    block kernel (PMBRL_DW_NO_WIDE) is the reference here -- complete and truncated horizons, one launch and
    the pipelined launches (where the later ranges ADD to the partial rows)."""
    d = common.load('c5_small')
    H = int(d['H'])
    assert H == 20
    os.environ['PMBRL_DW_NO_WIDE'] = '1'
    try:
        e0, g0, x0, a0 = run(d, 'off', True, n_valid=n_valid)
    finally:
        del os.environ['PMBRL_DW_NO_WIDE']
    e1, g1, x1, a1 = run(d, spec, True, n_valid=n_valid)
    assert e1.info['dw_pipe'] == (1 if spec == 'off' else 3)
    assert np.all(np.isfinite(g1))
    if n_valid == 0:
        assert np.all(g1 == 0.0) and np.all(g0 == 0.0)
    else:
        assert common.rel(g1, g0) < 2e-6
    if n_valid is None:
        assert common.rel(g1, d['ref64_grad']) < TOL_GRAD
