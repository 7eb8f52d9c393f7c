"""GPU (-m gpu): moment-matching groups spread over several ranks (pmbrl_config.mm_span_*), the ranks
as THREADS of one process: every rank holds a slice of every group, the per-step statistics cross the ranks through
pmbrl_plan_set_collective (here a barrier and a sum over the ranks' device buffers).  What the ranks compute
together must be what one process computes on all rows: the fixtures are runs of the real reference
(tools/make_golden.py) with its fp64 trajectory and policy gradient.  The multi-process form of the same path
(torch.distributed transport, mc_pilco) is in tests/test_gpu_two_ranks.py."""
import threading

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


class ThreadSum:
    """In-place sum over the ranks (threads) of a device tensor, same result on every rank."""

    def __init__(self, world):
        self.world, self.bar, self.views, self.total = world, threading.Barrier(world, timeout=30), [None] * world, None

    def rank(self, r):
        def allreduce(view):
            torch.cuda.synchronize()
            self.views[r] = view
            self.bar.wait()
            if r == 0:
                tot = self.views[0].clone()
                for v in self.views[1:]:
                    tot += v
                torch.cuda.synchronize()
                self.total = tot
            self.bar.wait()
            view.copy_(self.total)
            torch.cuda.synchronize()
            self.bar.wait()
        return allreduce


def _rows(B, G, parts, r):
    """Global rows of rank r: slice [off_r, off_r + parts[r]) of every group of M = B / G rows."""
    M = B // G
    off = sum(parts[:r])
    return np.concatenate([np.arange(g * M + off, g * M + off + parts[r]) for g in range(G)]), off


def _run(name, parts, precision=None, fail_at=None):
    from prob_mbrl_amd import rollout as RO
    d0 = common.load(name)
    B = d0['x0'].shape[0]
    G = max(int(d0['mm_groups']), 1)
    M, H, W = B // G, int(d0['H']), len(parts)
    assert sum(parts) == M
    ts = ThreadSum(W)
    res, err = [None] * W, []

    def worker(r):
        try:
            rows, off = _rows(B, G, parts, r)
            d = dict(d0)
            for k in list(d):
                if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
                    d[k] = d[k][rows]
            dyn, pol = common.modules_from_fixture(d, name, DEV)
            x0 = torch.tensor(d['x0'], device=DEV)
            # (the engine interface mc_pilco's fused path uses, not loss.backward(): autograd runs the device nodes
            #  of every thread's graph on ONE worker thread per device, so two thread-ranks would wait for each
            #  other inside it; ranks that are processes have an autograd engine each)
            bundle = RO.Bundle(dyn, pol, len(rows), H, False, False, True, True, G if G > 1 else None,
                               torch.tensor(d['z_mm'], device=DEV), torch.tensor(d['z_rr'], device=DEV),
                               B_global=B, precision=precision, mm_span=(M, off, W, r), process_group=ts.rank(r),
                               infer_ns=bool(d['infer_ns']) if 'infer_ns' in d else False)
            eng = bundle.engine
            S, A, R = bundle.forward(x0)
            n = min(eng.valid_steps(), H)
            if fail_at is not None:
                # a failure at step fail_at as the forward sweep reports it: status word, poisoned tail
                assert n == H
                n = fail_at
                eng.status[0] = n
                S[n + 1:], A[n:], R[n:] = float('nan'), float('nan'), float('nan')
            gw = torch.tensor(common.loss_weights(d, B)[:, :len(rows)], device=DEV)
            loss = float((R[:n, :, 0] * gw[:n]).sum())
            g, _, _ = eng.backward(gw)
            res[r] = (rows, S[:n + 1].cpu().numpy(), R[:n].cpu().numpy(), loss, g.double().cpu().numpy())
        except BaseException as e:   # noqa: BLE001
            err.append(e)
            ts.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(W)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=90)
    if err:
        raise err[0]
    n = res[0][1].shape[0]
    assert all(x[1].shape[0] == n for x in res)        # every rank kept the same horizon
    S = np.zeros((n, B, d0['x0'].shape[1]))
    for rows, s, _, _, _ in res:
        S[:, rows] = s
    return d0, S, sum(x[3] for x in res), sum(x[4] for x in res)


@pytest.mark.parametrize('name,parts', [
    ('mm1_d5', (12, 12)),              # one group of 24 rows over two ranks
    ('mm1_b100_h40', (40, 35, 25)),    # the H = 40 regime SURVEY 7 flags, unequal slices over three ranks
    ('mmg_d4', (6, 4)),                # four groups, each spread over two ranks
    ('mmg_h40', (15, 10)),
    ('dcp_d6_mmg', (5, 4, 3)),         # D = 6, three groups over three ranks
    ('full200_mmg', (13, 12)),         # the 200-wide cart-pole networks (latency-optimised kernel family)
    ('angles_dcp_mmg', (6, 6)),        # angle_dims inside Policy / DynamicsModel
    ('mmg_m80', (30, 50)),             # 80-row groups: two waves of a workgroup share a rank's rows
    ('mmg_infer_ns', (5, 3)),          # infer_noise_variables (utils/rollout.py:6-17): zhat = Delta L^-T
    ('c5_mm_d12', (20, 12)),           # D = 12: 12 x 12 statistics cross the ranks
    ('c5_mm_d32', (40, 24)),           # D = 32, 64-row groups (BASELINE.md's C5 widths)
])
def test_groups_spread_over_ranks_match_the_reference(name, parts):
    d, S, loss, grad = _run(name, parts)
    assert S.shape == d['ref64_states'].shape
    assert common.rel(S, d['ref64_states']) < 2e-5
    assert abs(loss - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
    assert common.rel(grad, d['ref64_grad']) < 1e-4


def test_groups_spread_over_ranks_in_exact_fp32():
    d, S, loss, grad = _run('mmg_d4', (5, 5), precision='f32')
    assert common.rel(S, d['ref64_states']) < 2e-5
    assert common.rel(grad, d['ref64_grad']) < 1e-4


def test_truncated_horizon_with_groups_spread_over_ranks():
    """utils/rollout.py:154-157 (a failure after more than 5 steps: the first n steps are kept): the adjoint's
    statistics exchange covers the completed steps only, every rank still issues every collective."""
    d0 = common.load('trunc_mm')
    d, S, loss, grad = _run('trunc_mm', (16, 14), fail_at=int(d0['fail_step']))
    assert S.shape == d['ref64_states'].shape
    assert common.rel(S, d['ref64_states']) < 2e-5
    assert abs(loss - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
    assert np.all(np.isfinite(grad)) and common.rel(grad, d['ref64_grad']) < 1e-4


def test_span_needs_a_collective_and_rejects_bad_slices():
    import prob_mbrl_amd as pm
    d = common.load('mmg_d4')
    rows, off = _rows(40, 4, (6, 4), 0)
    for k in list(d):
        if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
            d[k] = d[k][rows]
    dyn, pol = common.modules_from_fixture(d, 'mmg_d4', DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    kw = dict(resample_state_noise=False, resample_action_noise=False, mm_states=True, mm_rewards=True, mm_groups=4,
              z_mm=torch.tensor(d['z_mm'], device=DEV), z_rr=torch.tensor(d['z_rr'], device=DEV), B_global=40)
    with pytest.raises(ValueError):
        pm.utils.rollout(x0, dyn, pol, 4, mm_span=(10, 0, 2, 0), **kw)               # no process_group
    with pytest.raises(pm._lib.PmbrlError):
        pm.utils.rollout(x0, dyn, pol, 4, mm_span=(10, 5, 2, 1), process_group=lambda v: None, **kw)   # 5 + 6 > 10


def test_span_form_over_an_rccl_communicator_and_inside_a_graph():
    """pmbrl_plan_set_comm: the statistics exchange as ncclAllReduce (fp64, in place) on the compute stream, with the
    one rank a single-GPU box has (the sum over one rank is the identity: what this pins is the call itself --
    datatype, stream, buffer -- between the two halves of the moment matching, 2 H + 2 times per iteration), eager
    and recorded into a hipGraph; the result is what the in-kernel moment matching of the same rows gives."""
    import ctypes as C
    from prob_mbrl_amd import _lib, engine as E
    name = 'full200_mmg'
    d = dict(common.load(name))
    B, G = d['x0'].shape[0], int(d['mm_groups'])
    lib = _lib.load()
    idbuf = C.create_string_buffer(128)
    _lib.check(lib.pmbrl_comm_unique_id(idbuf), 'pmbrl_comm_unique_id')
    comm = C.c_void_p()
    _lib.check(lib.pmbrl_comm_init(C.c_char_p(bytes(idbuf.raw)), 0, 1, 0, C.byref(comm)), 'pmbrl_comm_init')
    gw = torch.tensor(common.loss_weights(d, B), device=DEV)
    eng0, args0, _ = common.engine_from_fixture(d, torch.device(DEV))
    S0, _, R0 = eng0.forward(**args0)
    g0 = eng0.backward(gw)[0].clone()
    eng, args, _ = common.engine_from_fixture(d, torch.device(DEV), mm_span=(B // G, 0, 1, 0))
    assert eng.info['mm_mode'] == 2
    _lib.check(lib.pmbrl_plan_set_comm(eng.plan, comm), 'pmbrl_plan_set_comm')
    S, _, R = eng.forward(**args)
    g = eng.backward(gw)[0].clone()
    assert eng.valid_steps() == eng.H
    assert common.rel(S.cpu().numpy(), S0.cpu().numpy()) < 1e-6 and common.rel(R.cpu().numpy(), R0.cpu().numpy()) < 1e-6
    assert common.rel(g.cpu().numpy(), g0.cpu().numpy()) < 1e-5
    assert common.rel(S.cpu().numpy(), d['ref64_states']) < 2e-5 and common.rel(g.cpu().numpy(), d['ref64_grad']) < 1e-4
    # the same launches and collectives recorded into one graph
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        eng.forward(**args)
        eng.backward(gw)
    torch.cuda.current_stream().wait_stream(s)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        eng.forward(**args)
        eng.backward(gw)
    eng.grad_flat.zero_()
    graph.replay()
    torch.cuda.synchronize()
    assert torch.equal(eng.grad_flat, g)
    del eng
    lib.pmbrl_comm_destroy(comm)


@pytest.mark.parametrize('world', [1, 2])
def test_sums_spread_over_workgroups_match_one_process(world):
    """Slices of more than 256 rows: the sums over a rank's rows are spread over several workgroups (a slot per
    workgroup forward, parts added in order in the adjoint).  The cart-pole shape with ONE group over all 2500 rows,
    as one process runs it (moment matching behind the device-wide barrier) and in the span form on 1 and 2 ranks."""
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0))
    d['mm_groups'] = 0
    d['H'] = 12
    B = d['x0'].shape[0]
    dev = torch.device(DEV)
    gw_all = torch.tensor(PB.loss_weights(d, B)[:12].copy(), device=DEV)
    eng0, args0, _ = PB.engine_from_problem(d, dev)
    S0, _, R0 = eng0.forward(**args0)
    g0 = eng0.backward(gw_all)[0].double().cpu().numpy()
    S0, R0 = S0.cpu().numpy(), R0.cpu().numpy()
    assert eng0.valid_steps() == 12
    ts = ThreadSum(world)
    res, err = [None] * world, []

    def worker(r):
        try:
            lo, hi = r * B // world, (r + 1) * B // world
            eng, args, _ = PB.engine_from_problem(d, dev, shard=(r, world), mm_span=(B, lo, world, r))
            assert eng.info['mm_mode'] == 2
            eng.attach_collective(ts.rank(r))
            S, _, R = eng.forward(**args)
            g = eng.backward(gw_all[:, lo:hi].contiguous())[0]
            res[r] = (lo, hi, S.cpu().numpy(), R.cpu().numpy(), g.double().cpu().numpy())
        except BaseException as e:   # noqa: BLE001
            err.append(e)
            ts.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=90)
    if err:
        raise err[0]
    for lo, hi, S, R, _ in res:
        assert common.rel(S, S0[:, lo:hi]) < 1e-5 and common.rel(R, R0[:, lo:hi]) < 1e-5
    assert common.rel(sum(x[4] for x in res), g0) < 1e-4


def test_one_process_group_beyond_the_cu_count_takes_the_one_rank_span_form(monkeypatch):
    """One group of 5000 rows: 313 workgroups cannot meet at a barrier under one launch, and the plan runs the
    span form with one rank (identity exchange, nothing to attach) instead of redoing the whole group's moment matching
    in the prologue of every workgroup of every step's launch.  Same trajectory and gradient as that per-step form."""
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0, P=200, H=8))
    d['mm_groups'] = 0
    B = d['x0'].shape[0]
    dev = torch.device(DEV)
    gw = torch.tensor(PB.loss_weights(d, B).copy(), device=DEV)
    monkeypatch.setenv('PMBRL_MM_PERSTEP', '1')
    eng0, args0, _ = PB.engine_from_problem(d, dev)
    assert eng0.info['mm_mode'] == 3 and eng0.info['mm_grid'] == 0
    S0, _, R0 = eng0.forward(**args0)
    g0 = eng0.backward(gw)[0].clone()
    monkeypatch.delenv('PMBRL_MM_PERSTEP')
    eng, args, _ = PB.engine_from_problem(d, dev)
    assert eng.info['mm_mode'] == 2
    S, _, R = eng.forward(**args)
    g = eng.backward(gw)[0]
    assert eng.valid_steps() == 8
    assert common.rel(S.cpu().numpy(), S0.cpu().numpy()) < 1e-5 and common.rel(R.cpu().numpy(), R0.cpu().numpy()) < 1e-5
    assert common.rel(g.cpu().numpy(), g0.cpu().numpy()) < 1e-4


@pytest.mark.parametrize('D,rows,world', [(16, 2304, 1), (16, 2304, 2), (32, 640, 1)])
def test_wide_state_sums_spread_over_workgroups(D, rows, world):
    """The span form at D >= 16 with enough rows per rank that the sums are spread over several workgroups (nb > 1):
    the adjoint's second half adds nb parts of d*d + d doubles each (staged behind the scratch, they
    overran the launch's LDS at D = 16 with >= 2048 rows, D = 32 with >= 512 rows).  One group over all rows,
    against the fp64 oracle and against the one-workgroup-per-group kernels (mm_mode 2 without span)."""
    from oracle import ref_torch as R
    from prob_mbrl_amd import problem as PB
    d = dict(PB.synthetic_problem('mid16_mm', seed=3, data_seed=0, P=4, S=rows // 4, H=4))
    if D == 32:
        d = dict(PB.synthetic_problem('stress32_mm', seed=3, data_seed=0, P=4, S=rows // 4, H=3))
    d['mm_groups'] = 0
    B, H = d['x0'].shape[0], int(d['H'])
    assert B == rows and d['x0'].shape[1] == D
    dev = torch.device(DEV)
    gw_all = torch.tensor(PB.loss_weights(d, B).copy(), device=DEV)
    x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
    torch.set_num_threads(16)
    l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, H, gamma, True, True, True, None, z_mm, z_rr)
    S64 = torch.stack(S64).detach().numpy()
    ts = ThreadSum(world)
    res, err = [None] * world, []

    def worker(r):
        try:
            lo, hi = r * B // world, (r + 1) * B // world
            eng, args, _ = PB.engine_from_problem(d, dev, shard=(r, world), mm_span=(B, lo, world, r))
            assert eng.info['mm_mode'] == 2
            eng.attach_collective(ts.rank(r))
            S, _, Rw = eng.forward(**args)
            assert eng.valid_steps() == H
            g = eng.backward(gw_all[:, lo:hi].contiguous())[0]
            res[r] = (lo, hi, S.cpu().numpy(), g.double().cpu().numpy())
        except BaseException as e:   # noqa: BLE001
            err.append(e)
            ts.bar.abort()

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=120)
    if err:
        raise err[0]
    for lo, hi, S, _ in res:
        assert common.rel(S, S64[:, lo:hi]) < 2e-5
    g = sum(x[3] for x in res)
    print('span D=%d rows=%d world=%d: grad %.2e' % (D, rows, world, common.rel(g, g64.numpy())))
    assert common.rel(g, g64.numpy()) < 1e-4
