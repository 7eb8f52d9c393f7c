"""CPU, gloo, world_size 2: the multi-GPU decomposition contract.  Each rank owns a
contiguous block of whole moment-matching groups (prob_mbrl_amd.distributed.shard_bounds),
computes its partial gradient -- here with the oracle's explicit adjoint standing in for
the HIP kernels, which need a GPU -- scaled by 1/B_global with the global cyclic noise
index, and ONE sum all-reduce reproduces the full-batch gradient.  The device-side twin
is tests/test_gpu_kernels.py::test_sharded_rows_reproduce_full_gradient."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from tests import common


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, name, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import adjoint_np as A
        from prob_mbrl_amd import distributed as D
        d = common.load(name)
        P = A.Problem(d, np.float64)
        B = P.x0.shape[0]
        G = int(d['mm_groups'])
        lo, hi = D.shard_bounds(B, G if G > 0 else None, world, rank)
        Q = P.shard(lo, hi, D.local_groups(B, G if G > 0 else None, world, rank))
        st = A.forward(Q)
        g, _, _ = A.backward(Q, st)
        t = torch.tensor(g)
        loss = torch.tensor([A.loss(Q, st)])
        D.allreduce_sum_(t)
        D.allreduce_sum_(loss)
        mx = D.max_over_ranks(float(rank), torch.device('cpu'))
        if rank == 0:
            out.put((t.numpy(), float(loss), mx, (lo, hi)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize('name', ['nomm_d4', 'mmg_d4', 'dcp_d6_mmg'])
def test_two_rank_shards_allreduce_to_full_gradient(name):
    world = 2
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, out)) for r in range(world)]
    for p in procs:
        p.start()
    g, loss, mx, bounds = out.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    d = common.load(name)
    assert common.rel(g, d['ref64_grad']) < 1e-6
    assert abs(loss - float(d['ref64_loss'])) <= 1e-7 * abs(float(d['ref64_loss']))
    assert mx == 1.0


def test_shard_bounds_cover_whole_groups():
    from prob_mbrl_amd import distributed as D
    for B, G, world in [(2500, 100, 8), (2500, None, 8), (40, 4, 3), (20000, 400, 4), (7, None, 2)]:
        seen = []
        for r in range(world):
            lo, hi = D.shard_bounds(B, G, world, r)
            if G:
                assert lo % (B // G) == 0 and hi % (B // G) == 0
            seen.append((lo, hi))
        assert seen[0][0] == 0 and seen[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(seen[:-1], seen[1:]))
