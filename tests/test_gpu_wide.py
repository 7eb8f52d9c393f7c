"""General family, 64-row workgroups with in-place layers on the 512-wide layers of pmbrl_wide.h (pm_rollout_fwd / _bwd
<4, 2, 2>): the 3 x 512 fixtures captured from the reference (c5_small: 16 rows -- a workgroup with 48 absent rows;
c5_mm_small: 128 rows in two workgroups, moment matching between per-step sweep launches), against the fp64 reference
outputs at the tolerances of every other form, and against the generic in-place form (PMBRL_WIDE=0) and the 16-row
two-buffer form of the same family: same stashes, same activity-bit layout, so the forward sweep of one form is also run
under the adjoint sweep of the other."""
import numpy as np
import pytest
import torch

from tests import common

pytestmark = pytest.mark.gpu

TOL_TRAJ = 2e-5
TOL_GRAD = 1e-4


@pytest.fixture(scope='module')
def dev():
    assert torch.cuda.is_available()
    return torch.device('cuda:0')


def _run(d, dev, hint, gw, **kw):
    eng, args, _ = common.engine_from_fixture(d, dev, rows_per_wg_hint=hint, force_generic=True, **kw)
    S, A, R = eng.forward(**args)
    g = eng.backward(gw)[0].cpu().numpy().copy()
    assert eng.valid_steps() == int(d['H'])
    return eng, S.cpu().numpy().copy(), A.cpu().numpy().copy(), R.cpu().numpy().copy(), g


@pytest.mark.parametrize('name', ['c5_small', 'c5_mm_small'])
def test_wide_layers_match_reference_and_other_forms(dev, name, monkeypatch):
    d = common.load(name)
    B = d['x0'].shape[0]
    gw = torch.tensor(common.loss_weights(d, B), device=dev)
    ew, Sw, Aw, Rw, gwide = _run(d, dev, 64, gw)
    assert ew.info['inplace'] == 2 and ew.info['rows_per_wg'] == 64 and ew.info['lds_bytes'] <= 160 * 1024
    e16, S16, A16, R16, g16 = _run(d, dev, 0, gw)
    assert e16.info['inplace'] == 0 and e16.info['rows_per_wg'] == 16
    monkeypatch.setenv('PMBRL_WIDE', '0')
    eg, Sg, Ag, Rg, gg = _run(d, dev, 64, gw)
    monkeypatch.delenv('PMBRL_WIDE')
    assert eg.info['inplace'] == 1
    for S, g in ((Sw, gwide), (S16, g16), (Sg, gg)):
        assert common.rel(S, d['ref64_states']) < TOL_TRAJ and common.rel(g, d['ref64_grad']) < TOL_GRAD
    assert common.rel(Aw, d['ref64_actions']) < TOL_TRAJ and common.rel(Rw, d['ref64_rewards']) < TOL_TRAJ
    # (hidden layers bit for bit the other forms'; the heads sum their K in another order than the two-buffer form's)
    assert common.rel(Sw, S16) < 2e-6 and common.rel(gwide, g16) < 2e-5
    assert common.rel(Sw, Sg) < 2e-6 and common.rel(gwide, gg) < 2e-5


def test_wide_is_the_default_when_every_cu_gets_64_rows(dev):
    """Plan choice: with B >= 64 x CU count and every hidden layer 512 wide the plan takes 64-row workgroups and the
    wide layers without being asked; below that it keeps the two-buffer form."""
    from prob_mbrl_amd import problem as PB
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    d = PB.synthetic_problem('stress32', seed=0, data_seed=0, P=cus, S=64, H=2)
    eng, args, _ = PB.engine_from_problem(d, dev)
    assert eng.info['inplace'] == 2 and eng.info['rows_per_wg'] == 64 and eng.info['n_wg'] == cus
    d = PB.synthetic_problem('stress32', seed=0, data_seed=0, P=cus // 2, S=64, H=2)
    eng, args, _ = PB.engine_from_problem(d, dev)
    assert eng.info['inplace'] == 0
