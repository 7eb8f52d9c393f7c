#!/usr/bin/env python
"""mm_groups=None -- ONE moment-matching group over the whole batch, the reference examples' default
(examples/deep_pilco_mm.py:31; utils/rollout.py:127-128) -- at the cart-pole shape, 2 500 rows x 40 steps:
forward + adjoint sweep per call pair (host wall clock over 20 pairs, the library's HIP-event timers beside it).
usage: bench_single_group.py [groups (0 = one group)]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prob_mbrl_amd import problem as PB  # noqa: E402


def main():
    groups = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    d = dict(PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0))
    d['mm_groups'] = np.asarray(groups)
    dev = torch.device('cuda:0')
    eng, args, _ = PB.engine_from_problem(d, dev)
    print(eng.info)
    B = d['x0'].shape[0]
    gw = torch.tensor(PB.loss_weights(d, B), device=dev)
    for _ in range(3):
        eng.forward(**args)
        eng.backward(gw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 20
    for _ in range(n):
        eng.forward(**args)
        eng.backward(gw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    eng.set_timing(True)
    acc = {}
    for _ in range(5):
        eng.forward(**args)
        eng.backward(gw)
        for k, v in eng.read_timing().items():
            acc[k] = acc.get(k, 0.0) + v / 5
    eng.set_timing(False)
    print('mm_groups=%s B=%d H=%d: %.3f ms per forward + adjoint call pair (wall) -> %.0f rollouts/s; valid %d; reg %s %s'
          % (groups or None, B, int(d['H']), dt * 1e3, B / dt, eng.valid_steps(), eng.info['reg'], eng.reg_calls()))
    print('   HIP-event timers (ms): ' + ', '.join('%s %.3f' % (k, v) for k, v in acc.items() if v > 0) +
          ' -> sweeps %.3f ms' % (acc.get('fwd', 0) + acc.get('bwd', 0)))


if __name__ == '__main__':
    main()
