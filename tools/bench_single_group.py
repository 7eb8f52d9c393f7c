import sys, time, numpy as np, torch
sys.path.insert(0, '/root/repo')
from prob_mbrl_amd import problem as PB, engine as E
d = dict(PB.synthetic_problem('cartpole_mm', seed=0, data_seed=0))
d['mm_groups'] = np.asarray(0)
dev = torch.device('cuda:0')
eng, args, _ = PB.engine_from_problem(d, dev)
print(eng.info)
B = d['x0'].shape[0]
gw = torch.tensor(PB.loss_weights(d, B), device=dev)
for _ in range(3):
    eng.forward(**args); eng.backward(gw)
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 20
for _ in range(n):
    eng.forward(**args); eng.backward(gw)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print('G=None B=2500 H=40: %.3f ms per fwd+bwd -> %.0f rollouts/s; valid %d' % (dt * 1e3, B / dt, eng.valid_steps()))
