"""GPU debugging aid: first-iteration policy gradient of a fixture, per parameter tensor, against the fp64 oracle --
the register-resident family and (PMBRL_REG=0) the latency-optimised one.   python tools/dbg/grad_by_layer.py <fixture>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402


def run(d, reg):
    os.environ['PMBRL_REG'] = '1' if reg else '0'
    dev = torch.device('cuda:0')
    eng, args, _ = common.engine_from_fixture(d, dev)
    S, A, R = eng.forward(**args)
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    g, _, _ = eng.backward(gw)
    torch.cuda.synchronize()
    g = g.cpu().numpy().copy()
    if reg:      # the other family's adjoint from the same stashes
        g2, _, _ = eng.backward(gw, want_x0=True, want_agn=True)
        torch.cuda.synchronize()
        run.g_mixed = g2.cpu().numpy().copy()
    return eng, S.cpu().numpy(), g


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'mcp_full200'
    d = common.load(name)
    from oracle import adjoint_np as ADJ
    P = ADJ.Problem(d, np.float64)
    st = ADJ.forward(P)
    g_ref, _, _ = ADJ.backward(P, st)
    S_ref = np.stack(st['states'])
    dims = [d['x0'].shape[1]] + [d['pol_W%d' % i].shape[0] for i in range(int(d['pol_n_layers']))]
    for reg in (True, False):
        eng, S, g = run(d, reg)
        print('reg=%d info reg %s calls %s | states rel %.2e grad rel %.2e, rms %.3e' %
              (reg, eng.info['reg'], eng.reg_calls(), common.rel(S, S_ref), common.rel(g, g_ref), np.sqrt(np.mean(g_ref ** 2))))
        if reg:
            print('   register-resident forward + latency-optimised adjoint: grad rel %.2e' % common.rel(run.g_mixed, g_ref))
        off = 0
        for l in range(len(dims) - 1):
            for kind, n in (('W', dims[l + 1] * dims[l]), ('b', dims[l + 1])):
                a, b = g[off:off + n], g_ref[off:off + n]
                e = np.abs(a - b)
                print('   %s%d: rel %.2e  max abs err %.2e (at |g| %.2e)  rms %.2e' %
                      (kind, l, common.rel(a, b), e.max(), abs(b[np.argmax(e)]), np.sqrt(np.mean(b ** 2))))
                off += n


if __name__ == '__main__':
    main()
