import sys, time, torch, numpy as np
sys.path.insert(0, '.')
from prob_mbrl_amd import problem as PB
dev = torch.device('cuda:0')
import os
for P, hint in ((80, 16), (60, 16), (40, 16)):
    pr = PB.synthetic_problem('dcartpole_mm', seed=0, data_seed=0, P=P)
    eng, args, _ = PB.engine_from_problem(pr, dev, rows_per_wg_hint=hint)
    B = pr['x0'].shape[0]
    gw = torch.tensor(PB.loss_weights(pr, B), device=dev)
    for _ in range(3):
        eng.forward(**args); eng.backward(gw)
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    n = 20
    tf = tb = 0.0
    for _ in range(n):
        e0.record(); eng.forward(**args); e1.record(); eng.backward(gw); e2.record()
        torch.cuda.synchronize()
        tf += e0.elapsed_time(e1); tb += e1.elapsed_time(e2)
    print('P=%d hint=%d B=%d' % (P, hint, B), {k: eng.info[k] for k in ('rows_per_wg', 'n_wg', 'row_tiles', 'mm_mode', 'mm_grid', 'dw_pipe')},
          'fwd %.3f ms bwd(+dW) %.3f ms  valid %d  -> %.0f rows/ms' % (tf / n, tb / n, eng.valid_steps(), B / ((tf + tb) / n)))
