"""Wall-clock of utils.train_regressor at the examples' shape (2000 iterations, minibatch 100)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import prob_mbrl_amd as pm
dev = 'cuda:0'
D, U, N = 4, 1, 1000
dyn = pm.models.DynamicsModel(
    pm.models.mlp(D + U, 2 * D, [200, 200], dropout_layers=[pm.models.CDropout(0.1 * np.ones(200)) for _ in range(2)], nonlin=torch.nn.ReLU),
    reward_func=None, output_density=pm.models.DiagGaussianDensity(D)).float()
dyn.set_dataset(torch.randn(N, D + U), 0.1 * torch.randn(N, D))
opt = torch.optim.Adam(dyn.parameters(), 1e-4)
dyn = dyn.to(dev)
pm.utils.train_regressor(dyn, 50, 100, True, opt)
torch.cuda.synchronize(); t0 = time.perf_counter()
pm.utils.train_regressor(dyn, 2000, 100, True, opt)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print('train_regressor: %.1f us / iteration (%.0f it/s)' % (dt / 2001 * 1e6, 2001 / dt))
