"""GPU (-m gpu), world_size 2 on ONE device: the N>1 control flow of bench.py and of
mc_pilco(process_group=...) with both ranks on cuda:0 and the collectives over gloo (RCCL refuses
two ranks on one GPU; the driver's multi-GPU bench is the RCCL run).  What is checked is what
cannot be checked on CPU: every rank issues the same sequence of collectives around real kernel
launches, and the sharded iteration reproduces the reference's single-process result."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu
ROOT = common.ROOT


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_bench_two_ranks_one_device():
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
           '--timing-steps', '2', '--dist-backend', 'gloo', '--one-device', '--repeats', '1']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['config']['global_rows'] == 2 * out['config']['rows_per_gpu']
    assert out['scaling'] == 'weak' and out['value'] > 0 and out['config']['debug_one_device']
    assert 'cpu_baseline' not in out
    # both curves in one line: the strong one is BASELINE.json's 100 x 25 rows divided over the ranks
    st = out['strong']
    assert st['global_rows'] == 2500 and st['rows_per_gpu'] == [1250, 1250] and st['value'] > 0
    assert 'rccl_ranks' in out and out['rccl_ranks'] is None      # gloo here: no RCCL communicator to ask
    # the shape of the N > 1 line: what a reader that keeps only the head of config.workload must still learn -- which
    # curve `value` is, the global row count, the other curve's value, what the communicator saw -- and whether the
    # transport fell back is a key of the parsed line whether it did or not
    head = out['config']['workload'][:200]
    assert head.startswith('N=2 WEAK: global_rows=5000 (2 x 2500 rows/GPU)'), head
    assert 'rccl_ranks=None' in head and 'transport=rccl' in head and 'STRONG curve (global_rows=2500) = ' in head, head
    assert 'transport_fallback' in out and out['transport_fallback'] is None and out['transport'] == 'rccl'


def test_bench_launches_itself_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment (how a driver that knows nothing of
    torch.distributed.run would start the scaling run): the script becomes its own launcher, rank 0 prints the one JSON
    line, and the line says which curve `value` is and what DESIGN.md predicted for it."""
    env = {k: v for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--timing-steps', '1',
           '--dist-backend', 'gloo', '--one-device', '--repeats', '2']
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out['n_gpus'] == 2 and out['scaling'] == 'weak' and out['timed_blocks'] == 2
    assert out['value_min'] <= out['value'] <= out['value_max']
    assert 'weak curve' in out['scaling_note'] or '`value` is the weak' in out['scaling_note']
    assert out['predicted']['weak']['value'] == 10.7e6 and out['strong']['global_rows'] == 2500


def test_bench_three_ranks_uneven_strong_split():
    """100 moment-matching groups over 3 ranks: 34 / 33 / 33 whole groups (strong curve), 3 x 2500 rows (weak)."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '3',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '3', '--steps', '2', '--warmup', '1', '--config', 'cartpole_mm',
           '--timing-steps', '1', '--dist-backend', 'gloo', '--one-device', '--repeats', '1']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['scaling'] == 'weak' and out['config']['global_rows'] == 7500
    assert out['strong']['rows_per_gpu'] == [850, 825, 825] and out['strong']['global_rows'] == 2500


def test_bench_two_ranks_strong_scaling():
    """--scaling strong: the SAME 100 x 25 rows divided over the ranks (what BASELINE.json's metric at
    N GPUs means); the JSON line says which it was."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1', '--scaling', 'strong',
           '--timing-steps', '2', '--dist-backend', 'gloo', '--one-device', '--repeats', '1']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['scaling'] == 'strong' and out['config']['global_rows'] == 2500 and out['config']['rows_per_gpu'] == 1250
    assert out['weak']['global_rows'] == 5000 and out['weak']['value'] > 0


def test_bench_two_ranks_one_global_moment_matching_group():
    """--mm-global: one moment-matching group over the rows of both ranks, statistics exchanged every step."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '2', '--warmup', '1', '--config', 'cartpole_mm',
           '--mm-global', '--timing-steps', '1', '--dist-backend', 'gloo', '--one-device', '--repeats', '1']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['config']['mm_mode'] == 2 and out['config']['global_rows'] == 5000 and out['value'] > 0


def _mcp_worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import prob_mbrl_amd as pm
        from prob_mbrl_amd import distributed as D
        d = dict(common.load(name))
        B = d['x0'].shape[0]
        lo, hi = D.shard_bounds(B, None, world, rank)
        for k in list(d):
            if k == 'x0' or k in ('pol_z', 'dyn_z') or '_mask' in k:
                d[k] = d[k][lo:hi]
        dyn, pol = common.modules_from_fixture(d, name, 'cuda:0')
        opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
        losses = []
        pm.algorithms.mc_pilco(
            torch.tensor(d['x0'], device='cuda:0'), dyn, pol, int(d['H']), opt, None,
            int(d['mcp_n_iters']), maximize=True, clip_grad=float(d['mcp_clip']),
            on_iteration=lambda i, loss, *a: losses.append(float(loss)),
            frozen_noise=(dict(z_mm=torch.tensor(d['z_mm']), z_rr=torch.tensor(d['z_rr'])) if bool(d['mm_states'])
                          else dict(z_mm=torch.zeros(1, 4), z_rr=torch.zeros(1, 1))),
            mm_states=bool(d['mm_states']), mm_rewards=bool(d['mm_rewards']),
            discount=None if np.allclose(d['gamma'], d['gamma'][0]) else float(d['gamma'][1] / d['gamma'][0]),
            cvar_eps=float(d['mcp_cvar_eps']) if 'mcp_cvar_eps' in d else 0.0,
            reg_weight=float(d['mcp_reg_weight']) if 'mcp_reg_weight' in d else 0.0,
            process_group=dist.group.WORLD)
        lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
        final = torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)])
        out.put((rank, losses, final.cpu().numpy()))
    finally:
        dist.destroy_process_group()


def test_mc_pilco_two_ranks_match_reference():
    """Rows sharded over two ranks (15 + 15 of the fixture's 30), gradient and loss all-reduced:
    the reference's single-process losses and final parameters on BOTH ranks."""
    import torch.multiprocessing as mp
    name = 'mcp_nomm'
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mcp_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    for rank, losses, final in res:
        assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
        assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    assert np.array_equal(res[0][2], res[1][2])      # replicas stay bit-identical


@pytest.mark.parametrize('world', [2, 3])
def test_mc_pilco_one_global_moment_matching_group_over_ranks(world):
    """mm_groups=None on a sharded run (the examples' default, examples/deep_pilco_mm.py:31): ONE Gaussian over the
    particles of all ranks at every step -- the ranks exchange the group's fp64 statistics per step, forward and in
    the adjoint (pmbrl_config.mm_span_rows).  The fixture is the REAL reference's single-process
    mc_pilco with moment-matched states and rewards over its 30 rows; here 15 + 15 and 10 + 10 + 10 rows on
    separate processes reproduce its losses and final parameters on every rank."""
    import torch.multiprocessing as mp
    name = 'mcp_mm1'
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mcp_worker, args=(r, world, port, name, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    for rank, losses, final in res:
        assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
        assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    for r in res[1:]:
        assert np.array_equal(res[0][2], r[2])      # replicas stay bit-identical


def _rollout_worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import prob_mbrl_amd as pm
        d = dict(common.load(name))
        B, H = d['x0'].shape[0], int(d['H'])
        lo, hi = rank * B // world, (rank + 1) * B // world
        for k in list(d):
            if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
                d[k] = d[k][lo:hi]
        dyn, pol = common.modules_from_fixture(d, name, 'cuda:0')
        x0 = torch.tensor(d['x0'], device='cuda:0')
        S, A, R = pm.utils.rollout(
            x0, dyn, pol, H, resample_state_noise=False, resample_action_noise=False, mm_states=True, mm_rewards=True,
            z_mm=torch.tensor(d['z_mm'], device='cuda:0'), z_rr=torch.tensor(d['z_rr'], device='cuda:0'),
            B_global=B, mm_span=(B, lo, world, rank), process_group=dist.group.WORLD)
        gamma = [float(g) for g in d['gamma']]
        loss = (-torch.stack([r * gamma[i] for i, r in enumerate(R)]).sum(0)).sum() / B
        pol.zero_grad()
        loss.backward()          # the adjoint's statistics exchange runs on autograd's device thread
        lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
        g = torch.cat([t.grad.reshape(-1) for l in lins for t in (l.weight, l.bias)]).double().cpu()
        tot = torch.cat([loss.detach().double().cpu().reshape(1), g])
        dist.all_reduce(tot)
        out.put((rank, lo, hi, torch.stack(S).detach().cpu().numpy(), float(tot[0]), tot[1:].numpy()))
    finally:
        dist.destroy_process_group()


def test_rollout_autograd_with_one_group_over_two_processes():
    """utils.rollout(..., mm_span=, process_group=) + loss.backward() on two processes, one moment-matching group of
    100 rows over both at H = 40: the reference's fp64 trajectory, loss and policy gradient (sum over the ranks)."""
    import torch.multiprocessing as mp
    name = 'mm1_b100_h40'
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rollout_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    S = np.zeros(d['ref64_states'].shape)
    for rank, lo, hi, s, loss, g in res:
        S[:, lo:hi] = s
        assert abs(loss - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
        assert common.rel(g, d['ref64_grad']) < 1e-4
    assert common.rel(S, d['ref64_states']) < 2e-5


@pytest.mark.parametrize('name', ['mcp_cvar', 'mcp_reg', 'mcp_cvar_neg_reg'])
def test_mc_pilco_two_ranks_cvar_and_regulariser(name):
    """CVaR (algorithms/mc_pilco.py:146-154: the quantile over the returns of ALL ranks' rows, gathered in row
    order) and the policy regulariser (:193-194) on a sharded run: the real reference's single-process losses and
    final parameters on both ranks."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mcp_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=120) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    for rank, losses, final in res:
        assert np.allclose(losses, d['ref32_mcp_losses'], rtol=5e-5)
        assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    assert np.array_equal(res[0][2], res[1][2])


def _n3_worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import prob_mbrl_amd as pm
        from prob_mbrl_amd import algorithms as ALG
        from tests.test_gpu_api import _value_from_fixture
        d = dict(common.load(name))
        B = d['x0'].shape[0]
        lo, hi = rank * B // world, (rank + 1) * B // world
        for k in list(d):
            if k == 'x0' or k in ('pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
                d[k] = d[k][lo:hi]
        dyn, pol = common.modules_from_fixture(d, name, 'cuda:0')
        opt = torch.optim.Adam(pol.parameters(), float(d['mcp_lr']))
        losses, x0s = [], []
        kw = dict(maximize=True, clip_grad=float(d['mcp_clip']), on_iteration=lambda i, loss, *a: losses.append(float(loss)),
                  frozen_noise=dict(z_mm=torch.zeros(1, 4), z_rr=torch.zeros(1, 1)), process_group=dist.group.WORLD)
        exp = None
        if name == 'ext_value':
            full = dict(common.load(name))
            V = _value_from_fixture(full)
            # the critic's frozen masks / noise are per row: this rank's rows of them
            for m in V.modules():
                for attr in ('noise', 'concrete_noise', 'z'):
                    t = getattr(m, attr, None)
                    if isinstance(t, torch.Tensor) and t.dim() == 2 and t.shape[0] == B:
                        if attr == 'concrete_noise':
                            m.concrete_noise = t[lo:hi].clone()
                        else:
                            getattr(m, attr).data = t[lo:hi].clone()
            kw['value_func'] = V
        else:
            exp = pm.utils.ExperienceDataset()
            for e in range(int(d['replay_n_episodes'])):
                st = d['replay_states%d' % e]
                T = len(st)
                exp.append_episode(list(st), list(np.zeros((T, 1), np.float32)), list(np.zeros(T)), [None] * T, None)
            ALG.x0_tree, ALG.episode_counter = None, 0
            np.random.seed(int(d['replay_np_seed']))          # the same numpy stream on every rank
            kw.update(prioritized_replay=True, priority_alpha=0.6, init_priority_beta=0.4, priority_beta_increase=0.1,
                      on_rollout=lambda i, s, a, r, disc: x0s.append(s[0].detach().cpu().numpy()))
        pm.algorithms.mc_pilco(torch.tensor(d['x0'], device='cuda:0'), dyn, pol, int(d['H']), opt, exp,
                               int(d['mcp_n_iters']), **kw)
        lins = [m for m in pol.model._modules.values() if isinstance(m, torch.nn.Linear)]
        final = torch.cat([t.detach().reshape(-1) for l in lins for t in (l.weight, l.bias)]).cpu().numpy()
        extra = None
        if exp is not None:
            tree = ALG.x0_tree
            n = len(d['replay_final_counts'])
            extra = (np.stack(x0s), tree.counts[:n].copy(), tree.sum_tree[tree.max_size - 1:tree.max_size - 1 + n].copy())
        out.put((rank, lo, hi, losses, final, extra))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['ext_value', 'ext_replay'])
def test_mc_pilco_two_ranks_value_bootstrap_and_prioritised_replay(name):
    """mc_pilco(value_func=V) and mc_pilco(prioritized_replay=True) on rows sharded over two processes
    (algorithms/mc_pilco.py:136-140,156-188,222-246): the terminal value and its gradient are per row; the priority tree
    is replicated -- every rank draws the same global sample of start states, keeps its slice, and updates the whole
    sample's priorities from the gathered ||dL/da_t||.  Both reproduce the real reference's single-process run."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_n3_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([out.get(timeout=180) for _ in range(2)], key=lambda x: x[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    for rank, lo, hi, losses, final, extra in res:
        assert np.allclose(losses, d['ref32_mcp_losses'], rtol=1e-4)
        assert np.allclose(final, d['ref32_mcp_final'], rtol=1e-4, atol=2e-6)
    assert np.allclose(res[0][4], res[1][4], rtol=0, atol=0)        # replicas stay identical
    if name == 'ext_replay':
        x0s = np.concatenate([res[0][5][0], res[1][5][0]], axis=1)   # [iters, B, D]: the two slices side by side
        assert np.array_equal(x0s, d['replay_x0s'].astype(np.float32))
        for r in res:
            assert np.array_equal(r[5][1], d['replay_final_counts'])
            assert np.allclose(r[5][2], d['replay_final_leaves'], rtol=1e-3)


def test_bench_two_ranks_p2p_transport_one_device():
    """bench.py --transport p2p with both ranks on one device: the gradient all-reduce of every iteration really runs on
    the device between two processes (IPC-mapped slots), which RCCL cannot do on a one-GPU box."""
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
           '--master-addr', '127.0.0.1', '--master-port', str(_free_port()),
           os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '5', '--warmup', '2', '--transport', 'p2p',
           '--timing-steps', '2', '--dist-backend', 'gloo', '--one-device', '--repeats', '1', '--no-second-curve']
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][0])
    assert out['n_gpus'] == 2 and out['value'] > 0 and 'peer-to-peer' in out['collective']


def _agree_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        import prob_mbrl_amd as pm
        from prob_mbrl_amd import distributed as D
        from prob_mbrl_amd import engine as E
        d = dict(common.load('nomm_d4'))
        B, H = d['x0'].shape[0], int(d['H'])
        lo, hi = D.shard_bounds(B, None, world, rank)
        for k in list(d):
            if k == 'x0' or k in ('pol_z', 'dyn_z') or '_mask' in k:
                d[k] = d[k][lo:hi]
        dyn, pol = common.modules_from_fixture(d, 'nomm_d4', 'cuda:0')
        real = E.Engine.valid_steps
        if rank == 1:      # this rank's device "finds" a failure at step 8 of 12: the other rank must stop there too
            def failing(self):
                real(self)
                self.status[0] = 8
                return 8
            E.Engine.valid_steps = failing
        S, A, R = pm.utils.rollout(torch.tensor(d['x0'], device='cuda:0'), dyn, pol, H, resample_state_noise=False,
                                   resample_action_noise=False, agree_group=dist.group.WORLD, B_global=B, row_offset=lo)
        E.Engine.valid_steps = real
        # p2p_check without a cached peer-to-peer object on one of the ranks: both must still enter its all-reduce
        os.environ['PMBRL_P2P'] = '1'
        if rank == 0:
            D._P2PS[(id(None), 'cuda:0')] = type('FakeP2P', (), {'failed': lambda self: False})()
        D.p2p_check(None, torch.device('cuda:0'))
        del os.environ['PMBRL_P2P']
        D._P2PS.clear()
        out.put((rank, len(S), len(A), len(R)))
    finally:
        dist.destroy_process_group()


def test_ranks_agree_on_a_truncated_horizon_and_all_enter_p2p_check():
    """rollout(agree_group=...): a failure only ONE rank's device reports truncates the horizon on every rank (a rank
    that truncated alone would leave the others inside the next collective); p2p_check enters its all-reduce on every
    rank whether or not the rank holds a cached peer-to-peer object (a rank that returned early hung the others)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(out.get(timeout=300) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res == [(0, 9, 8, 8), (1, 9, 8, 8)], res
