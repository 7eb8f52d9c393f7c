"""GPU (-m gpu): the one-shot peer-to-peer all-reduce (pmbrl_p2p_*, csrc/pmbrl_p2p.hip) with the ranks as two / three
PROCESSES on one device -- IPC-mapped buffers work where RCCL refuses two ranks per GPU, so this is also the first
run of the statistics exchange of moment-matching groups spread over ranks through a device-side transport with more
than one rank.  Bootstrap (the 64-byte IPC handles) over a gloo group."""
import os
import socket

import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _allreduce_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from prob_mbrl_amd.distributed import P2PComm
        comm = P2PComm(None, 'cuda:0', max_bytes=1 << 20)
        res = []
        for it, (n, dt) in enumerate([(1, torch.float64), (1057, torch.float64), (41602, torch.float32), (7, torch.float32),
                                      (200000, torch.float32), (131072, torch.float64), (41602, torch.float32)] * 2):
            g = torch.Generator().manual_seed(1000 * it + rank)
            x = torch.randn(n, generator=g, dtype=dt)
            t = x.to('cuda:0')
            comm.allreduce_(t)
            res.append(t.cpu().numpy())
        torch.cuda.synchronize()
        assert not comm.failed()
        out.put((rank, res))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('world', [2, 3])
def test_p2p_allreduce_between_processes_on_one_device(world):
    """Sums of random fp32 / fp64 vectors (1 .. 200 000 elements, 14 calls: both slot sets, several generations): what
    every rank ends up with is the sum in rank order -- bit-identical on all ranks and equal to the host's."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_allreduce_worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    res = dict(out.get(timeout=180) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sizes = [(1, torch.float64), (1057, torch.float64), (41602, torch.float32), (7, torch.float32),
             (200000, torch.float32), (131072, torch.float64), (41602, torch.float32)] * 2
    for it, (n, dt) in enumerate(sizes):
        want = None
        for r in range(world):
            x = torch.randn(n, generator=torch.Generator().manual_seed(1000 * it + r), dtype=dt)
            want = x.clone() if want is None else want + x        # rank order, the device's order
        for r in range(world):
            assert np.array_equal(res[r][it], want.numpy()), (it, n, r)


def _span_worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from prob_mbrl_amd import problem as PB
        from prob_mbrl_amd.distributed import P2PComm
        d = dict(common.load(name))
        B, H = d['x0'].shape[0], int(d['H'])
        G = max(int(d['mm_groups']), 1)
        M = B // G
        parts = [M // world + (1 if r < M % world else 0) for r in range(world)]      # unequal slices allowed
        lo_in, per = sum(parts[:rank]), parts[rank]
        rows = np.concatenate([np.arange(g * M + lo_in, g * M + lo_in + per) for g in range(G)])
        for k in list(d):
            if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
                d[k] = d[k][rows]
        comm = P2PComm(None, 'cuda:0', max_bytes=1 << 20)
        eng, args, _ = PB.engine_from_problem(d, 'cuda:0', B_global=B, row_offset=0, mm_span=(M, lo_in, world, rank))
        eng.attach_collective(comm)
        S, _, R = eng.forward(**args)
        assert eng.valid_steps() == H
        gw = torch.tensor(common.loss_weights(d, B)[:, :len(rows)], device='cuda:0')
        g = eng.backward(gw)[0]
        tot = torch.cat([g.double(), (R[:, :, 0] * gw).sum().double().reshape(1)])
        comm.allreduce_(tot)
        torch.cuda.synchronize()
        assert not comm.failed()
        out.put((rank, rows, S.cpu().numpy(), tot[:-1].cpu().numpy(), float(tot[-1])))
        dist.barrier()
        comm.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['mmg_h40', 'c5_mm_d32'])
def test_groups_spread_over_two_processes_through_the_p2p_transport(name):
    """Moment-matching groups spread over two process-ranks, the per-step fp64 statistics (forward and adjoint) crossing
    through pmbrl_p2p (attached like any transport: Engine.attach_collective): the reference's fp64 trajectory, loss and
    policy gradient (fixtures of the real reference: four 25-row groups at H = 40; two 64-row groups at D = 32)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_span_worker, args=(r, 2, port, name, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in range(2)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    S = np.zeros(d['ref64_states'].shape)
    for rank, rows, s, g, loss in res:
        S[:, rows] = s
        assert abs(loss - float(d['ref64_loss'])) <= 2e-5 * abs(float(d['ref64_loss']))
        assert common.rel(g, d['ref64_grad']) < 1e-4
    assert common.rel(S, d['ref64_states']) < 2e-5
    assert np.array_equal(res[0][3], res[1][3])
