#!/usr/bin/env python
"""What the one-shot peer-to-peer all-reduce (pmbrl_p2p.hip) costs per call, with the ranks as PROCESSES on one
device (the only multi-rank arrangement a one-GPU box offers; between GPUs the stores cross xGMI instead of the local
fabric): message sizes of the per-step statistics exchange (4 groups x 29 doubles ... 256 groups x 1121 doubles) and of
the flat policy gradient (41 602 / 550 416 floats), against a host-staged gloo all-reduce of the same tensors; then
the cart-pole shape with ONE moment-matching group over the rows of both ranks (bench.py --mm-global's arrangement)
per forward + adjoint, statistics through p2p and through gloo.

    python tools/p2p_cost.py [world]
"""
import os
import socket
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from prob_mbrl_amd import problem as PB
    from prob_mbrl_amd.distributed import P2PComm
    dev = torch.device('cuda:0')
    comm = P2PComm(None, dev, max_bytes=4 << 20)
    lines = []
    for n, dt, what in ((4 * 29, torch.float64, 'statistics, 4 groups D=4'), (100 * 29, torch.float64, 'statistics, 100 groups D=4'),
                        (256 * 1121, torch.float64, 'statistics, 256 groups D=32'), (41602, torch.float32, 'gradient, 2x200 policy'),
                        (550416, torch.float32, 'gradient, 3x512 policy')):
        t = torch.randn(n, dtype=dt, device=dev)
        for _ in range(20):
            comm.allreduce_(t)
        torch.cuda.synchronize()
        dist.barrier()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200):
            comm.allreduce_(t)
        e1.record()
        torch.cuda.synchronize()
        us_p2p = e0.elapsed_time(e1) * 1e3 / 200
        dist.barrier()
        t0 = time.perf_counter()
        for _ in range(50):
            h = t.cpu()
            dist.all_reduce(h)
            t.copy_(h)
        torch.cuda.synchronize()
        us_gloo = (time.perf_counter() - t0) * 1e6 / 50
        lines.append('%-32s %8d B: p2p %7.1f us   host-staged gloo %8.1f us' % (what, n * t.element_size(), us_p2p, us_gloo))
    # one moment-matching group over the rows of all ranks, cart-pole shape
    d = dict(PB.synthetic_problem('cartpole_mm', seed=0, data_seed=rank))
    d['mm_groups'] = 0
    B, H = d['x0'].shape[0], int(d['H'])
    Bg = B * world
    gen = np.random.default_rng(12345)
    d['z_mm'] = gen.standard_normal((H + Bg, d['x0'].shape[1])).astype(np.float32)
    d['z_rr'] = gen.standard_normal((H + Bg, 1)).astype(np.float32)
    for mode in ('p2p', 'gloo'):
        eng, args, _ = PB.engine_from_problem(d, dev, B_global=Bg, row_offset=rank * B, mm_span=(Bg, rank * B, world, rank))
        eng.attach_collective(comm if mode == 'p2p' else dist.group.WORLD)
        gw = torch.tensor(PB.loss_weights(d, Bg)[:, :B].copy(), device=dev)
        for _ in range(3):
            eng.forward(**args)
            eng.backward(gw)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        n_it = 10
        for _ in range(n_it):
            eng.forward(**args)
            eng.backward(gw)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) * 1e3 / n_it
        assert eng.valid_steps() == H
        lines.append('one %d-row group over %d ranks, H=%d (2 H + 2 = %d exchanges): %.2f ms per forward + adjoint, statistics through %s'
                     % (Bg, world, H, 2 * H + 2, ms, mode))
        del eng
    assert not comm.failed()
    if rank == 0:
        out.put(lines)
    dist.barrier()
    comm.close()
    dist.destroy_process_group()


def main():
    import torch.multiprocessing as mp
    world = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    ctx = mp.get_context('spawn')
    out = ctx.Queue()
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    procs = [ctx.Process(target=worker, args=(r, world, port, out)) for r in range(world)]
    for p in procs:
        p.start()
    lines = out.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
    print('%d process-ranks on ONE MI355X; IPC-mapped uncached buffers; per call, steady state' % world)
    for l in lines:
        print(l)


if __name__ == '__main__':
    main()
