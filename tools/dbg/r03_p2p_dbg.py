import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import torch.multiprocessing as mp
from tests import common

def worker(rank, world, port, name, mode, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'; os.environ['MASTER_PORT'] = str(port)
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from prob_mbrl_amd import problem as PB
    from prob_mbrl_amd.distributed import P2PComm
    d = dict(common.load(name))
    B, H = d['x0'].shape[0], int(d['H'])
    G = max(int(d['mm_groups']), 1); M = B // G; per = M // world; lo_in = rank * per
    rows = np.concatenate([np.arange(g * M + lo_in, g * M + lo_in + per) for g in range(G)])
    for k in list(d):
        if k in ('x0', 'pol_z', 'dyn_z') or ('_mask' in k and not k.endswith(('_shape', '_bits'))):
            d[k] = d[k][rows]
    eng, args, _ = PB.engine_from_problem(d, 'cuda:0', B_global=B, row_offset=0, mm_span=(M, lo_in, world, rank))
    comm = None
    if mode == 'p2p':
        comm = P2PComm(None, 'cuda:0', max_bytes=1 << 20)
        eng.attach_collective(comm)
    else:
        eng.attach_collective(dist.group.WORLD)
    S, _, R = eng.forward(**args)
    gw = torch.tensor(common.loss_weights(d, B)[:, :len(rows)], device='cuda:0')
    loss = float((R[:, :, 0] * gw).sum())
    g = eng.backward(gw)[0]
    torch.cuda.synchronize()
    Sd = S.cpu().numpy()
    errs = [common.rel(Sd[t], d['ref64_states'][t][rows]) for t in range(0, H + 1, 5)]
    out.put((rank, mode, eng.valid_steps(), loss, errs, comm.failed() if comm else None))
    dist.barrier()
    dist.destroy_process_group()

if __name__ == '__main__':
    ctx = mp.get_context('spawn')
    for mode in ('gloo', 'p2p'):
        out = ctx.Queue()
        import socket
        s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
        procs = [ctx.Process(target=worker, args=(r, 2, port, 'mmg_h40', mode, out)) for r in range(2)]
        for p in procs: p.start()
        for _ in range(2):
            r = out.get(timeout=120)
            print(r[0], r[1], 'n', r[2], 'loss', r[3], 'errs', ' '.join('%.1e' % e for e in r[4]), 'failed', r[5], flush=True)
        for p in procs: p.join(timeout=60)
