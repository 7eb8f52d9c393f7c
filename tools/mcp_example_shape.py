"""mc_pilco at the reference examples' own shape (100 particles, H=15, one moment-matching group)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import prob_mbrl_amd as pm
dev = 'cuda:0'
D, U = 4, 1
dyn = pm.models.DynamicsModel(
    pm.models.mlp(D + U, 2 * D, [200, 200], dropout_layers=[pm.models.CDropout(0.1 * np.ones(200)) for _ in range(2)], nonlin=torch.nn.ReLU),
    reward_func=pm.rewards.CartpoleReward(pole_length=torch.tensor(0.5)), output_density=pm.models.DiagGaussianDensity(D)).float()
pol = pm.models.Policy(pm.models.mlp(D, 2 * U, [200, 200], dropout_layers=[pm.models.BDropout(0.1) for _ in range(2)], nonlin=torch.nn.ReLU,
                       output_nonlin=partial(pm.models.DiagGaussianDensity, U)), np.array([10.0], np.float32), np.array([-10.0], np.float32)).float()
dyn.set_dataset(torch.randn(300, D + U), 0.01 * torch.randn(300, D))
dyn, pol = dyn.to(dev), pol.to(dev)
opt = torch.optim.Adam(pol.parameters(), 1e-3)
x0 = 0.1 * torch.randn(100, D, device=dev)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
pm.algorithms.mc_pilco(x0, dyn, pol, 15, opt, None, 10, mm_states=True, mm_rewards=True)
torch.cuda.synchronize(); t0 = time.perf_counter()
pm.algorithms.mc_pilco(x0, dyn, pol, 15, opt, None, n, mm_states=True, mm_rewards=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('%.3f ms / iteration' % (dt / n * 1e3))
