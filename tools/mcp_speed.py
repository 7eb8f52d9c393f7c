import time, sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
from functools import partial
import prob_mbrl_amd as pm
dev = 'cuda:0'
D, U, H = 4, 1, 40
def build():
    dyn = pm.models.DynamicsModel(
        pm.models.mlp(D + U, 2 * D, [200, 200], dropout_layers=[pm.models.CDropout(0.1 * np.ones(200)) for _ in range(2)], nonlin=torch.nn.ReLU),
        reward_func=pm.rewards.CartpoleReward(pole_length=torch.tensor(0.5)), output_density=pm.models.DiagGaussianDensity(D)).float()
    pol = pm.models.Policy(pm.models.mlp(D, 2 * U, [200, 200], dropout_layers=[pm.models.BDropout(0.1) for _ in range(2)], nonlin=torch.nn.ReLU,
                           output_nonlin=partial(pm.models.DiagGaussianDensity, U)), np.array([10.0], np.float32), np.array([-10.0], np.float32)).float()
    dyn.set_dataset(torch.randn(300, D + U), 0.01 * torch.randn(300, D))
    return dyn.to(dev), pol.to(dev)
for mm, B, G in ((False, 2500, None), (True, 2500, 100)):
    dyn, pol = build()
    opt = torch.optim.Adam(pol.parameters(), 1e-4)
    x0 = 0.1 * torch.randn(B if G is None else G, D, device=dev)
    if G is not None:
        x0 = x0  # tiled inside mc_pilco
    kw = dict(mm_states=mm, mm_rewards=mm, mm_groups=G)
    init = torch.zeros(B, D, device=dev) if G is None else x0
    # mc_pilco wants init_states with N_particles rows; with mm_groups the [G, D] states are tiled
    init_states = (0.1 * torch.randn(B, D, device=dev))
    pm.algorithms.mc_pilco(init_states, dyn, pol, H, opt, None, 20, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    n = 300
    pm.algorithms.mc_pilco(init_states, dyn, pol, H, opt, None, n, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('mm=%s: %.3f ms / mc_pilco iteration (%.0f rollouts/s)' % (mm, dt / n * 1e3, B * n / dt))

# the reference examples' own shape: 100 particles, H=15, one moment-matching group over all rows
dyn, pol = build()
opt = torch.optim.Adam(pol.parameters(), 1e-3)
x0 = 0.1 * torch.randn(100, D, device=dev)
for mm in (True, False):
    kw = dict(mm_states=mm, mm_rewards=mm, mm_groups=None)
    pm.algorithms.mc_pilco(x0, dyn, pol, 15, opt, None, 20, **kw)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pm.algorithms.mc_pilco(x0, dyn, pol, 15, opt, None, 300, **kw)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print('example shape (B=100, H=15, mm=%s): %.3f ms / mc_pilco iteration' % (mm, dt / 300 * 1e3))
