"""GPU debugging aid: cycle stamps of pm_bnn_fwd_bwd's phases (PMBRL_BNN_PROF=1).  python tools/dbg/bnn_phases.py"""
import os
import sys

import numpy as np
import torch

os.environ['PMBRL_BNN_PROF'] = '1'
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from prob_mbrl_amd import engine as E  # noqa: E402

dev = torch.device('cuda:0')
h, N, M = 200, 1000, 100
dims = [5, h, h, 8]
torch.manual_seed(0)
parts = []
for l in range(3):
    parts += [torch.randn(dims[l + 1], dims[l], device=dev).reshape(-1) / np.sqrt(dims[l]), torch.zeros(dims[l + 1], device=dev)]
    if l < 2:
        parts.append(torch.full((dims[l + 1],), 1.1, device=dev))
flat = torch.cat(parts).contiguous()
step = E.BnnStep(dims, [0.1, 0.1], [0.5, 0.5], [1.0, 1.0], M, N, 1.0, device=dev)
Xn, Yn = torch.randn(N, 5, device=dev), torch.randn(N, 4, device=dev)
m, v, sc = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros(1, dtype=torch.int64, device=dev)
idx = torch.randint(0, N, (20, M), device=dev, dtype=torch.int32)
step.train_steps(Xn, Yn, idx, flat, m, v, sc, 1e-4, seed=1)
torch.cuda.synchronize()
ws = step.ws.cpu().numpy()
st = ws[len(ws) - 2048:len(ws) - 2048 + 21 * 8].view(np.int64)      # (the scratch region is the workspace's last 4 KB)
names = {0: 'entry', 1: 'gather', 2: 'fwd L0', 3: 'fwd L1', 10: 'head', 11: 'nll', 12: 'bwd head', 13: 'bwd L1', 20: 'end'}
prev = None
for k in sorted(names):
    if prev is not None:
        print('%-10s %8d cycles' % (names[k], st[k] - st[prev]))
    prev = k
print('total %d cycles (%.1f us at 2.3 GHz)' % (st[20] - st[0], (st[20] - st[0]) / 2300.0))
