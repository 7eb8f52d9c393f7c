"""GPU debugging aid: register-resident family vs the latency-optimised one on a synthetic config at chosen sizes;
first step at which the states disagree.   python tools/dbg/reg_vs_fast.py config [P [S [H]]]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from prob_mbrl_amd import problem as PB  # noqa: E402


def run(d, reg):
    os.environ['PMBRL_REG'] = '1' if reg else '0'
    dev = torch.device('cuda:0')
    eng, args, _ = PB.engine_from_problem(d, dev)
    gw = torch.tensor(PB.loss_weights(d, d['x0'].shape[0]), device=dev)
    S, A, R = eng.forward(**args)
    g, _, _ = eng.backward(gw)
    torch.cuda.synchronize()
    return eng, S.cpu().numpy().copy(), g.cpu().numpy().copy()


def main():
    cfg = sys.argv[1]
    P = int(sys.argv[2]) if len(sys.argv) > 2 else None
    S = int(sys.argv[3]) if len(sys.argv) > 3 else None
    H = int(sys.argv[4]) if len(sys.argv) > 4 else None
    d = dict(PB.synthetic_problem(cfg, seed=0, data_seed=0, P=P, S=S, H=H))
    e1, S1, g1 = run(d, True)
    e0, S0, g0 = run(d, False)
    print('info new', {k: e1.info[k] for k in ('rows_per_wg', 'n_wg', 'mm_parts', 'reg', 'mm_mode')}, 'valid', e1.valid_steps(), e0.valid_steps())
    rel = lambda a, b: float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))
    print('per-step states rel:', ' '.join('%.1e' % rel(S1[t], S0[t]) for t in range(0, S0.shape[0], max(1, S0.shape[0] // 30))))
    M = int(d['x0'].shape[0] // max(1, int(d['mm_groups'])))
    tt = max(0, e1.valid_steps() - 1)
    print('per-group states rel at step %d:' % tt, ' '.join('%.1e' % rel(S1[tt, g * M:(g + 1) * M], S0[tt, g * M:(g + 1) * M]) for g in range(min(16, S0.shape[1] // M))))
    for t in range(S0.shape[0]):
        r = rel(S1[t], S0[t])
        if r > 1e-5 or t == S0.shape[0] - 1:
            bad = np.argwhere(np.abs(S1[t] - S0[t]) > 1e-4 * (np.abs(S0[t]).max() + 1e-30))
            print('step %d: states rel %.2e; rows off: %s' % (t, r, sorted(set(int(b[0]) for b in bad))[:24]))
            break
    print('grad rel %.2e' % rel(g1, g0))


if __name__ == '__main__' and not (len(sys.argv) > 1 and sys.argv[1] == 'ratios'):
    main()


def ratios():
    """PR_MM_DEBUG_RATIO build: the smallest pivot / diagonal ratio of every (step, group) from the factor records"""
    import re
    import subprocess
    cfg, P, S, H = sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
    d = dict(PB.synthetic_problem(cfg, seed=0, data_seed=0, P=P, S=S, H=H))
    os.environ['PMBRL_REG'] = '1'
    os.environ['PMBRL_REG_DEBUG'] = '1'
    dev = torch.device('cuda:0')
    eng, args, _ = PB.engine_from_problem(d, dev)
    eng.forward(**args)
    torch.cuda.synchronize()
    off = int(os.environ['OFF_MMFAC'])
    D = d['x0'].shape[1]
    fd = 5 * D + D * D
    a0 = (-eng.workspace.data_ptr()) % 256
    ws = eng.workspace[a0 + off:a0 + off + H * P * fd * 8].cpu().numpy().view(np.float64).reshape(H, P, fd)
    r = ws[:, :, 3 * D]
    np.set_printoptions(linewidth=250, precision=1)
    print('valid', eng.valid_steps())
    for g in range(min(P, 8)):
        print('group %d min pivot ratio per step:' % g, ' '.join('%.0e' % v for v in r[:, g]))


if len(sys.argv) > 1 and sys.argv[1] == 'ratios':
    ratios()
