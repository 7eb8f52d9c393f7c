import sys, hashlib, numpy as np, torch
sys.path.insert(0, '/root/repo')
from tests import common
for name in ('full200_mmg', 'dcp200_mmg50', 'mmg_h40'):
    d = common.load(name)
    dev = torch.device('cuda:0')
    eng, args, _ = common.engine_from_fixture(d, dev)
    S, A, R = eng.forward(**args)
    gw = torch.tensor(common.loss_weights(d, d['x0'].shape[0]), device=dev)
    g, _, _ = eng.backward(gw)
    torch.cuda.synchronize()
    print(name, hashlib.md5(R.cpu().numpy().tobytes()).hexdigest()[:12], hashlib.md5(g.cpu().numpy().tobytes()).hexdigest()[:12])
