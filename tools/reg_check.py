"""GPU: the register-resident sweep family (pmbrl_reg.h) against the latency-optimised one (PMBRL_REG=0) on the
cart-pole shape at full size: trajectories, actions, gradient; then timings of both.

    python tools/reg_check.py [config]
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from prob_mbrl_amd import problem as PB  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def run(d, reg):
    os.environ['PMBRL_REG'] = '1' if reg else '0'
    dev = torch.device('cuda:0')
    eng, args, _ = PB.engine_from_problem(d, dev)
    B = d['x0'].shape[0]
    gw = torch.tensor(PB.loss_weights(d, B), device=dev)
    S, A, R = eng.forward(**args)
    g, _, _ = eng.backward(gw)
    torch.cuda.synchronize()
    out = (S.cpu().numpy().copy(), A.cpu().numpy().copy(), R.cpu().numpy().copy(), g.cpu().numpy().copy(), eng.valid_steps())
    # timing
    eng.set_timing(True)
    acc = {}
    for _ in range(10):
        eng.forward(**args)
        eng.backward(gw)
        for k, ms in eng.read_timing().items():
            if ms >= 0:
                acc.setdefault(k, []).append(ms)
    eng.set_timing(False)
    return out, {k: float(np.mean(v)) for k, v in acc.items()}, eng


def main():
    config = sys.argv[1] if len(sys.argv) > 1 else 'cartpole_nomm'
    d = dict(PB.synthetic_problem(config, seed=0, data_seed=0))
    (S0, A0, R0, g0, n0), t0, _ = run(d, False)
    (S1, A1, R1, g1, n1), t1, eng = run(d, True)
    print('valid steps: old %d new %d' % (n0, n1))
    print('states rel %.3e  actions rel %.3e  rewards rel %.3e  grad rel %.3e' % (rel(S1, S0), rel(A1, A0), rel(R1, R0), rel(g1, g0)))
    print('finite:', np.isfinite(S1).all(), np.isfinite(g1).all())
    print('old kernel ms:', {k: round(v, 4) for k, v in t0.items()})
    print('new kernel ms:', {k: round(v, 4) for k, v in t1.items()})
    if len(sys.argv) > 2 and sys.argv[2] == 'oracle':
        from oracle import ref_torch as R
        x0, pol, dyn, spec, meta, z_mm, z_rr, gamma = R.problem_from_npz(d, torch.float64)
        torch.set_num_threads(16)
        l64, g64, (S64, A64, R64) = R.iteration(x0, pol, dyn, spec, meta['H'], gamma, True, meta['mm_states'],
                                                meta['mm_rewards'], meta['mm_groups'], z_mm, z_rr)
        S64 = torch.stack(S64).detach().numpy()
        print('vs fp64 oracle: states old %.3e new %.3e | grad old %.3e new %.3e' %
              (rel(S0, S64), rel(S1, S64), rel(g0, g64.numpy()), rel(g1, g64.numpy())))


if __name__ == '__main__':
    main()
