"""GPU debugging aid: forward sweep of a fixture on the register-resident family and on the latency-optimised one; which
regions of the workspace (the stashes the adjoint reads) differ?   python tools/dbg/ws_diff.py <fixture>"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402


def fwd(d, reg):
    os.environ['PMBRL_REG'] = '1' if reg else '0'
    os.environ['PMBRL_REG_DEBUG'] = '1'
    dev = torch.device('cuda:0')
    eng, args, _ = common.engine_from_fixture(d, dev)
    eng.workspace.zero_()
    S, A, R = eng.forward(**args)
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    torch.cuda.synchronize()
    off = (-eng.workspace.data_ptr()) % 256
    g, _, _ = eng.backward(gw, want_x0=True, want_agn=True)      # the latency-optimised adjoint in both cases
    torch.cuda.synchronize()
    ws = eng.workspace[off:].cpu().numpy().copy()      # (behind the adjoint call: weights repacked, activity bits unpacked)
    return ws, A.cpu().numpy().copy(), g.cpu().numpy().copy()


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'mcp_full200'
    d = common.load(name)
    w1, A1, g1 = fwd(d, True)
    w0, A0, g0 = fwd(d, False)
    n = min(len(w0), len(w1)) // 4 * 4
    f0, f1 = w0[:n].view(np.float32), w1[:n].view(np.float32)
    print('actions rel %.2e  grad (same adjoint) rel %.2e' % (common.rel(A1, A0), common.rel(g1, g0)))
    # exact byte comparison (activity bits are nibble bytes: as floats they are denormals no tolerance sees)
    nb = np.nonzero(w0[:n] != w1[:n])[0]
    nb = nb[nb < 7020032] if name.startswith('mcp_full200') else nb[:0]
    print('bytes that differ in front of the stashes: %d' % len(nb), nb[:20], [(int(w0[i]), int(w1[i])) for i in nb[:20]])
    CH = 1024
    bad = []
    for i in range(0, len(f0), CH):
        a, b = f0[i:i + CH].astype(np.float64), f1[i:i + CH].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        if not fin.all() or np.abs(a - b).max() > 1e-5 * max(np.abs(a).max(), 1e-30):
            bad.append(i)
    # contiguous ranges
    rng = []
    for i in bad:
        if rng and i == rng[-1][1]:
            rng[-1][1] = i + CH
        else:
            rng.append([i, i + CH])
    for lo, hi in rng[:40]:
        a, b = f0[lo:hi].astype(np.float64), f1[lo:hi].astype(np.float64)
        fin = np.isfinite(a) & np.isfinite(b)
        print('bytes [%d, %d): max abs diff %.3e, max |ref| %.3e, rel L2 %.2e' %
              (lo * 4, hi * 4, np.abs(a - b)[fin].max() if fin.any() else -1, np.abs(a[fin]).max() if fin.any() else -1,
               common.rel(b[fin], a[fin])))


if __name__ == '__main__':
    main()
