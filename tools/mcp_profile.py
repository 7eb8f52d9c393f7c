"""cProfile of the host side of mc_pilco's fused iteration (C2 shape)."""
import cProfile
import pstats
import sys
import os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from functools import partial
import prob_mbrl_amd as pm

dev = 'cuda:0'
D, U, H, B = 4, 1, 40, 2500
dyn = pm.models.DynamicsModel(
    pm.models.mlp(D + U, 2 * D, [200, 200], dropout_layers=[pm.models.CDropout(0.1 * np.ones(200)) for _ in range(2)], nonlin=torch.nn.ReLU),
    reward_func=pm.rewards.CartpoleReward(pole_length=torch.tensor(0.5)), output_density=pm.models.DiagGaussianDensity(D)).float()
pol = pm.models.Policy(pm.models.mlp(D, 2 * U, [200, 200], dropout_layers=[pm.models.BDropout(0.1) for _ in range(2)], nonlin=torch.nn.ReLU,
                       output_nonlin=partial(pm.models.DiagGaussianDensity, U)), np.array([10.0], np.float32), np.array([-10.0], np.float32)).float()
dyn.set_dataset(torch.randn(300, D + U), 0.01 * torch.randn(300, D))
dyn, pol = dyn.to(dev), pol.to(dev)
opt = torch.optim.Adam(pol.parameters(), 1e-4)
x0 = 0.1 * torch.randn(B, D, device=dev)
pm.algorithms.mc_pilco(x0, dyn, pol, H, opt, None, 20)
pr = cProfile.Profile()
pr.enable()
pm.algorithms.mc_pilco(x0, dyn, pol, H, opt, None, 300)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
