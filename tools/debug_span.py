"""Debug: moment-matching groups 'spread' over ONE rank (identity collective) against the in-kernel path."""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from tests import common
import prob_mbrl_amd as pm

name = sys.argv[1] if len(sys.argv) > 1 else 'mcp_mm1'
d = common.load(name)
DEV = 'cuda:0'
H = int(d['H'])
B = d['x0'].shape[0]
out = {}
for mode in ('plain', 'span'):
    dyn, pol = common.modules_from_fixture(d, name, DEV)
    x0 = torch.tensor(d['x0'], device=DEV)
    kw = dict(mm_states=True, mm_rewards=True, mm_groups=None, z_mm=torch.tensor(d['z_mm'], device=DEV),
              z_rr=torch.tensor(d['z_rr'], device=DEV))
    if mode == 'span':
        kw.update(mm_span=(B, 0, 1, 0), process_group=lambda v: None)
    S, A, R = pm.utils.rollout(x0, dyn, pol, H, resample_state_noise=False, resample_action_noise=False, **kw)
    loss = -torch.stack(R).sum(0).mean()
    pol.zero_grad()
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in pol.parameters()])
    out[mode] = (torch.stack(S).detach().cpu().numpy(), torch.stack(R).detach().cpu().numpy(), g.cpu().numpy())
    print(mode, float(loss))
for i, nm in enumerate(('states', 'rewards', 'grad')):
    a, b = out['plain'][i], out['span'][i]
    print(nm, 'rel', np.linalg.norm(a - b) / np.linalg.norm(a))
    if nm != 'grad':
        for t in range(min(4, a.shape[0])):
            print('  t', t, np.linalg.norm(a[t] - b[t]) / (np.linalg.norm(a[t]) + 1e-30))
