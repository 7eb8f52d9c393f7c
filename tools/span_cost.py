"""What the per-step statistics exchange of moment-matching groups spread over ranks costs on ONE device, without
the transport: the cartpole_mm shape with ONE group over all 2500 rows (mm_groups=None, the examples' default)
 (a) as one process runs it (moment matching inside the sweeps / behind a device-wide barrier), and
 (b) in the form a sharded run uses (pmbrl_config.mm_span_*: one sweep launch per step, statistics kernel ->
     collective -> factor-and-apply kernel, forward and adjoint) with a collective that does nothing (one rank).
(b) - (a) is the launch structure's price; (c) the same with the statistics going through ncclAllReduce on a
one-rank RCCL communicator (the launches of the real transport, none of its link latency), (d) recorded into a
hipGraph and replayed.
    python tools/span_cost.py [config] [iterations]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prob_mbrl_amd import problem as PB  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else 'cartpole_mm'
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
# optional third argument: particles P (rows = P x 25): the size of the one group
d = dict(PB.synthetic_problem(name, seed=0, data_seed=0, P=int(sys.argv[3]) if len(sys.argv) > 3 else None))
d['mm_groups'] = 0
B, H = d['x0'].shape[0], int(d['H'])
import ctypes as C  # noqa: E402
from prob_mbrl_amd import _lib  # noqa: E402
lib = _lib.load()
idbuf = C.create_string_buffer(128)
_lib.check(lib.pmbrl_comm_unique_id(idbuf), 'pmbrl_comm_unique_id')
comm = C.c_void_p()
_lib.check(lib.pmbrl_comm_init(C.c_char_p(bytes(idbuf.raw)), 0, 1, 0, C.byref(comm)), 'pmbrl_comm_init')
for label, span in (('one process', None), ('span form, 1 rank, no transport', (B, 0, 1, 0)),
                    ('span form, 1 rank, RCCL all-reduce', (B, 0, 1, 0)),
                    ('span form, 1 rank, RCCL, hipGraph', (B, 0, 1, 0))):
    eng, args, _ = PB.engine_from_problem(d, dev, mm_span=span)
    if span and 'RCCL' in label:
        # the real transport with the one rank this box has: 2 H + 2 ncclAllReduce launches per iteration
        _lib.check(lib.pmbrl_plan_set_comm(eng.plan, comm), 'pmbrl_plan_set_comm')
    elif span:
        eng.attach_collective(lambda view: None)
    gw = torch.tensor(PB.loss_weights(d, B).copy(), device=dev)

    def step():
        eng.forward(**args)
        eng.backward(gw)
    for _ in range(3):
        step()
    if 'hipGraph' in label:
        torch.cuda.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            eng.forward(**args)
            eng.backward(gw)
        step = graph.replay       # noqa: F811
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / iters * 1e3
    print('%-34s mm_mode %d  mm_grid %d  valid steps %d / %d  %.3f ms per forward + adjoint  (%.0f k rollouts/s)' %
          (label, eng.info['mm_mode'], eng.info['mm_grid'], eng.valid_steps(), H, ms, B / ms))
