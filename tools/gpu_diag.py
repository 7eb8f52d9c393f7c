#!/usr/bin/env python
"""GPU diagnostic: per-step / per-tensor error of the HIP path vs the golden
fixtures (run on the GPU box; prints, never asserts)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import common  # noqa: E402


def main():
    names = sys.argv[1:] or common.fixture_names('iter')
    dev = torch.device('cuda:0')
    for name in names:
        d = common.load(name)
        if bool(d.get('infer_ns', False)):
            continue
        try:
            eng, args, _ = common.engine_from_fixture(d, dev)
            S, A, Rw = eng.forward(**args)
            B = d['x0'].shape[0]
            gw = torch.tensor(common.loss_weights(d, B), device=dev)
            loss = float(eng.weighted_sum(Rw, gw))
            g, gx0, agn = eng.backward(gw, want_x0=True, want_agn=True)
            torch.cuda.synchronize()
        except Exception as e:  # noqa: BLE001
            print('%-20s EXCEPTION %r' % (name, e))
            continue
        S, A, Rw, g = S.cpu().numpy(), A.cpu().numpy(), Rw.cpu().numpy(), g.cpu().numpy()
        print('== %s info=%s valid=%d' % (name, eng.info, eng.valid_steps()))
        H = int(d['H'])
        es = [common.rel(S[t], d['ref64_states'][t]) for t in range(H + 1)]
        ea = [common.rel(A[t], d['ref64_actions'][t]) for t in range(H)]
        er = [common.rel(Rw[t].reshape(-1), d['ref64_rewards'][t].reshape(-1)) for t in range(H)]
        print('   states  rel/step: ' + ' '.join('%.1e' % e for e in es[:8]) + ' ... max %.1e' % max(es))
        print('   actions rel/step: ' + ' '.join('%.1e' % e for e in ea[:8]) + ' ... max %.1e' % max(ea))
        print('   rewards rel/step: ' + ' '.join('%.1e' % e for e in er[:8]) + ' ... max %.1e' % max(er))
        print('   loss %.9g ref64 %.9g ref32 %.9g' % (loss, float(d['ref64_loss']), float(d['ref32_loss'])))
        off = 0
        for i in range(int(d['pol_n_layers'])):
            for nm in ('W', 'b'):
                n = d['pol_%s%d' % (nm, i)].size
                print('   grad %s%d rel %.2e (|ref| %.2e)' %
                      (nm, i, common.rel(g[off:off + n], d['ref64_grad'][off:off + n]),
                       np.linalg.norm(d['ref64_grad'][off:off + n])))
                off += n
        print('   grad total rel vs ref64 %.2e ; ref32 vs ref64 %.2e' %
              (common.rel(g, d['ref64_grad']), common.rel(d['ref32_grad'], d['ref64_grad'])))


if __name__ == '__main__':
    main()
