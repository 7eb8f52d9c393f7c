"""GPU debugging aid: the moment matching's factor records of a fixture, register-resident forward vs the latency-optimised one."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests import common  # noqa: E402


def fwd(d, reg):
    os.environ['PMBRL_REG'] = '1' if reg else '0'
    dev = torch.device('cuda:0')
    eng, args, _ = common.engine_from_fixture(d, dev)
    eng.workspace.zero_()
    eng.forward(**args)
    torch.cuda.synchronize()
    off = (-eng.workspace.data_ptr()) % 256
    return eng.workspace[off:].cpu().numpy().copy(), eng


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else 'full200_mmg'
    off = int(sys.argv[2]) if len(sys.argv) > 2 else 9223680
    D = int(sys.argv[3]) if len(sys.argv) > 3 else 4
    d = common.load(name)
    w1, e1 = fwd(d, True)
    w0, e0 = fwd(d, False)
    print(e1.info)
    n = 5 * D + D * D
    G = e1.info.get('mm_groups', 0) or 1
    f1 = w1[off:off + 8 * n * 400].view(np.float64)
    f0 = w0[off:off + 8 * n * 400].view(np.float64)
    np.set_printoptions(precision=5, linewidth=200)
    for rec in range(0, 6):
        print('record', rec)
        print(' fast', f0[rec * n:(rec + 1) * n])
        print(' reg ', f1[rec * n:(rec + 1) * n])


main()
