import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from tests import common
from prob_mbrl_amd import problem as PB
DEV = torch.device('cuda:0')
for name in ('c5_small', 'c5_mm_small', 'c5_mm_d32', 'rdv_d8_u4_3layer'):
    d = common.load(name)
    B = d['x0'].shape[0]
    for hint in (0, 64):
        eng, args, _ = common.engine_from_fixture(d, DEV, rows_per_wg_hint=hint, force_generic=True)
        S, A, R = eng.forward(**args)
        gw = torch.tensor(common.loss_weights(d, B), device=DEV)
        g = eng.backward(gw)[0].cpu().numpy()
        print(name, 'hint', hint, 'rows/wg', eng.info['rows_per_wg'], 'rt', eng.info['row_tiles'], 'n', eng.valid_steps(),
              'states %.2e actions %.2e grad %.2e' % (common.rel(S.cpu().numpy(), d['ref64_states']), common.rel(A.cpu().numpy(), d['ref64_actions']), common.rel(g, d['ref64_grad'])), flush=True)
