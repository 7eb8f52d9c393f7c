#!/usr/bin/env python
"""In-kernel phase profile of workgroup 0 (shader-clock stamps, pmbrl_plan_set_prof).
usage: phase_prof.py [config [rows_per_wg_hint [particles samples [mm_groups]]]]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from prob_mbrl_amd import _lib, problem as PB  # noqa: E402


def main():
    cfg = sys.argv[1] if len(sys.argv) > 1 else 'cartpole_nomm'
    hint = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    dev = torch.device('cuda:0')
    # optional: particles, samples, moment-matching groups (0 = one group over all rows)
    P = int(sys.argv[3]) if len(sys.argv) > 3 else None
    S = int(sys.argv[4]) if len(sys.argv) > 4 else None
    d = PB.synthetic_problem(cfg, seed=0, data_seed=0, P=P, S=S)
    if len(sys.argv) > 5:
        d['mm_groups'] = np.asarray(int(sys.argv[5]))
    eng, args, _ = PB.engine_from_problem(d, dev, rows_per_wg_hint=hint)
    print(eng.info)
    H = eng.H
    pf = torch.zeros(H * 32, dtype=torch.int64, device=dev)
    pb = torch.zeros(H * 32, dtype=torch.int64, device=dev)
    gw = torch.tensor(PB.loss_weights(d, eng.B), device=dev)
    for it in range(3):
        eng.forward(**args)
        eng.backward(gw)
    _lib.check(eng.lib.pmbrl_plan_set_prof(eng.plan, C.c_void_p(pf.data_ptr()), C.c_void_p(pb.data_ptr())), 'prof')
    eng.forward(**args)
    eng.backward(gw)
    torch.cuda.synchronize()
    for name, buf in (('fwd', pf), ('bwd', pb)):
        a = buf.cpu().numpy().reshape(H, 32)
        print('== %s: cycles between marks, median over steps (slot: delta)' % name)
        slots = [s for s in range(32) if a[H // 2, s] != 0]
        if not slots:
            print('   (no stamps: this variant carries no cycle stamps)')
            continue
        order = sorted(slots, key=lambda s: a[H // 2, s])
        prev = None
        tot = 0
        for s in order:
            if prev is not None:
                dl = np.median(a[1:-1, s] - a[1:-1, prev])
                tot += dl
                print('   %2d -> %2d : %8.0f cycles' % (prev, s, dl))
            prev = s
        if a[0, 30] and a[0, 31]:      # whole-launch stamps (register-resident family)
            first = a[0, 0] if name == 'fwd' else a[H - 1, 0]
            last = a[H - 1, 5] if name == 'fwd' else a[0, 5]
            cyc, rt = a[0, 31] - a[0, 30], (a[0, 29] - a[0, 28]) / 100.0      # cycles, microseconds
            print('   launch (workgroup 0): %d cycles = %.1f us -> %.2f GHz; prologue %d cycles, epilogue %d' %
                  (cyc, rt, cyc / rt / 1e3 if rt else 0.0, first - a[0, 30], a[0, 31] - last))
            if a[0, 27]:
                print('   prologue stamps (cycles from entry):', [int(a[0, k] - a[0, 30]) for k in (27, 26, 25) if a[0, k]])
        # partner workgroup's chain stations (slots 16..20 = its 8..12), relative to ITS OWN first stamp -- clocks of
        # different CUs are not comparable
        if a[H // 2, 16] != 0:
            ps = [s for s in range(16, 24) if a[H // 2, s] != 0]
            print('   partner workgroup (slots %s): deltas %s' % (ps, [int(np.median(a[1:-1, ps[i + 1]] - a[1:-1, ps[i]])) for i in range(len(ps) - 1)]))
        print('   step total (marked span): %.0f cycles; step period: %.0f' %
              (tot, np.median(np.abs(np.diff(a[:, order[0]])))))


if __name__ == '__main__':
    main()
