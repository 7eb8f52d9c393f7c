"""Host data path: ExperienceDataset / SumTree / checkpoints against vectors
captured from the reference (tools/make_golden.py: experience_host)."""
import ast
import os

import numpy as np
import torch

from tests import common

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load():
    return np.load(os.path.join(common.GOLDEN, 'experience_host.npz'), allow_pickle=False)


def _dataset(d):
    from prob_mbrl_amd.utils import ExperienceDataset
    exp = ExperienceDataset()
    for e in range(int(d['n_episodes'])):
        S, A, R = d['S%d' % e], d['A%d' % e], d['R%d' % e]
        exp.append_episode([s for s in S], [a for a in A], [r for r in R], dones=[False] * len(S),
                           infos=[{}] * len(S), ts=list(range(len(S))))
    return exp


def test_get_dynmodel_dataset_matches_reference():
    d = _load()
    exp = _dataset(d)
    assert exp.n_episodes() == 3 and exp.n_samples() == 24
    for k in range(int(d['n_opts'])):
        opts = dict(ast.literal_eval(str(d['opt%d' % k])))
        X, Y = exp.get_dynmodel_dataset(**opts)
        assert X.dtype == torch.float64 and Y.dtype == torch.float64
        assert X.shape == d['X%d' % k].shape and Y.shape == d['Y%d' % k].shape, (opts, X.shape, d['X%d' % k].shape)
        assert np.array_equal(X.numpy(), d['X%d' % k]), opts
        assert np.array_equal(Y.numpy(), d['Y%d' % k]), opts


def test_sum_tree_matches_reference():
    from prob_mbrl_amd.utils import SumTree
    d = _load()
    tree = SumTree(16)
    for i, p in enumerate(d['tree_pri']):
        tree.append(i * 10, p)
    tree.renormalize()
    np.random.seed(3 + 1)
    for k, (bs, beta) in enumerate(((4, 0.4), (40, 1.0), (8, 0.7))):
        samples, idxs, w = tree.sample(bs, beta=beta)
        assert np.array_equal(np.asarray(samples), d['tree_samples%d' % k])
        assert np.array_equal(np.asarray(idxs), d['tree_idxs%d' % k])
        assert np.allclose(np.asarray(w), d['tree_w%d' % k], rtol=1e-12, atol=0)
        tree.update(int(idxs[0]), 0.37)
        tree.renormalize()
    assert np.allclose(tree.sum_tree, d['tree_sum_tree'], rtol=1e-12, atol=1e-15)
    assert np.array_equal(tree.counts, d['tree_counts'])
    assert np.allclose([tree.idx, tree.max_p, tree.max_count, tree.size, tree.norm_factor], d['tree_scalars'])
    idx, p, data = tree.get(0.5 * tree.sum_tree[0])
    assert data == tree.data[idx - tree.max_size + 1] and p == tree.sum_tree[idx]


def test_experience_roundtrip_sampling_and_checkpoint(tmp_path):
    from prob_mbrl_amd.utils import ExperienceDataset, load_checkpoint
    d = _load()
    exp = _dataset(d)
    f = str(tmp_path / 'ckpt' / 'experience.pth.tar')
    exp.save(f)
    sd = torch.load(f, weights_only=False)
    assert set(sd) == {'states', 'actions', 'rewards', 'info', 'done', 'time_stamps', 'curr_episode',
                       'policy_parameters'}      # the reference's on-disk keys
    exp2 = ExperienceDataset()
    exp2.load(f)
    assert exp2.n_samples() == exp.n_samples() and exp2.curr_episode == exp.curr_episode
    np.random.seed(0)
    x0 = exp2.sample_states(50, timestep=0)
    assert x0.shape == (50, 4) and x0.dtype == torch.float64
    firsts = np.stack([d['S%d' % e][0] for e in range(3)])
    assert all(any(np.array_equal(r, f0) for f0 in firsts) for r in x0.numpy())
    assert exp2.sample_states(5, timestep=None).shape == (5, 4)
    # add_sample on a fresh dataset; truncate / reset
    exp3 = ExperienceDataset()
    for t in range(4):
        exp3.add_sample(np.zeros(4) + t, np.ones(2), 0.5, False, {}, t)
    assert exp3.n_episodes() == 1 and exp3.n_samples() == 4
    exp2.truncate(1)
    assert exp2.n_episodes() == 2
    exp2.reset()
    assert exp2.n_samples() == 0 and exp2.curr_episode == -1

    class _M:
        def load(self, sd):
            self.sd = sd
    dyn, pol = _M(), _M()
    torch.save({'a': torch.ones(2)}, str(tmp_path / 'ckpt' / 'latest_policy.pth.tar'))
    exp4 = ExperienceDataset()
    import warnings
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter('always')
        load_checkpoint(str(tmp_path / 'ckpt'), dyn, pol, exp4)
    assert hasattr(pol, 'sd') and not hasattr(dyn, 'sd') and exp4.n_samples() == 24
    assert any('latest_dynamics' in str(x.message) for x in w)


def test_apply_controller_loop():
    from prob_mbrl_amd.utils import apply_controller

    class Env:
        dt = 0.1

        def reset(self):
            self.t = 0
            return np.zeros(3)

        def step(self, u):
            self.t += 1
            return np.full(3, self.t, dtype=float), float(u.sum()), self.t >= 5, {}

    seen = []
    states, actions, costs, dones, infos = apply_controller(
        Env(), lambda x, t=None: np.array([[1.0, 2.0]]), 20, callback=lambda *a: seen.append(a))
    assert len(states) == 5 and dones[-1] and costs[0] == 3.0 and len(seen) == 5
    assert np.array_equal(states[3], np.full(3, 3.0)) and actions[0].shape == (2,)


def test_set_dataset_normalises_the_expanded_input():
    """models/core.py:136-146: with angle_dims the normalisation statistics are those of
    to_complex(X) = [others | sin | cos] (utils/angles.py:39-42), which is what the rollout's dynamics
    input is built from (fixtures angles_*)."""
    import prob_mbrl_amd as pm
    d = common.load('angles_dcp_mmg')
    D, U = d['x0'].shape[1], d['pol_z'].shape[1]
    ad = [int(a) for a in d['dyn_angle_dims']]
    dyn = pm.models.DynamicsModel(pm.models.mlp(D + U + len(ad), 2 * D, [8]), angle_dims=ad,
                                  output_density=pm.models.DiagGaussianDensity(D))
    g = torch.Generator().manual_seed(0)
    X, Y = torch.randn(50, D + U, generator=g), torch.randn(50, D, generator=g)
    dyn.set_dataset(X, Y)
    others = [i for i in range(D + U) if i not in ad]
    Xe = torch.cat([X[:, others], X[:, ad].sin(), X[:, ad].cos()], -1)
    assert dyn.X.shape == (50, D + U + len(ad))
    assert torch.allclose(dyn.mx, Xe.mean(0, keepdim=True))
    assert torch.allclose(dyn.iSx, (4.0 * Xe.std(0, keepdim=True)).reciprocal())
    assert d['dyn_mx'].shape[0] == D + U + len(ad)


def test_prob_mbrl_import_alias():
    """compat/prob_mbrl: reference scripts' `from prob_mbrl import utils, models, algorithms` without an edit."""
    import subprocess
    import sys
    code = ('import sys; sys.path[:0] = [%r, %r]; '
            'from prob_mbrl import utils, models, algorithms; import prob_mbrl.models as m; '
            'import prob_mbrl_amd.models as a; assert m.DynamicsModel is a.DynamicsModel and '
            'm.modules.CDropout is a.CDropout and hasattr(utils, "rollout") and '
            'hasattr(algorithms, "mc_pilco") and hasattr(models, "DynamicsModel")'
            % (os.path.join(ROOT, 'compat'), ROOT))
    assert subprocess.run([sys.executable, '-c', code]).returncode == 0


def test_dropout_draws_match_the_reference():
    """models/modules.py:40-58 (BDropout.update_noise) and :95-118 (CDropout.update_noise /
    update_concrete_noise): with the reference's seed the modules here draw the same uniforms, hand the same
    concrete probabilities to torch.bernoulli and keep the same hard sample (fixture draw_dropout: the
    reference's own calls, recorded by tools/make_golden.py)."""
    import torch
    from prob_mbrl_amd import models as M
    d = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'draw_dropout.npz'))
    B, h, seed = int(d['B']), int(d['h']), int(d['seed'])
    for k in range(2):
        cd = M.CDropout(torch.tensor(d['rate%d' % k], dtype=torch.float32), temperature=float(d['temp%d' % k]))
        cd.eval()
        assert np.allclose(cd.logit_p.detach().numpy(), d['logit_p%d' % k], rtol=1e-6, atol=1e-7)
        seen = []
        orig = torch.bernoulli

        def bern(p, *a, **kw):
            out = orig(p, *a, **kw)
            seen.append(p.detach().clone())
            return out

        torch.bernoulli = bern
        try:
            cd.update_noise(torch.empty(B, h), seed=seed + k)
        finally:
            torch.bernoulli = orig
        assert np.array_equal(cd.noise.numpy().astype(np.float32), d['u%d' % k].astype(np.float32))
        assert len(seen) == 1 and np.allclose(seen[0].numpy(), d['probs%d' % k], rtol=1e-5, atol=1e-7)
        hard = cd.concrete_noise.detach().numpy()
        assert set(np.unique(hard)) <= {0.0, 1.0}
        assert np.array_equal(hard.astype(np.uint8), d['hard%d' % k])
        assert np.allclose(cd.p.detach().numpy(), d['p_after%d' % k], rtol=1e-6)
        assert cd.keep_prob() == 1.0          # eval mode: x * mask, no division (models/modules.py:158-160)
    bd = M.BDropout(torch.tensor(d['b_rate'], dtype=torch.float32))
    bd.update_noise(torch.empty(B, h), seed=seed + 7)
    assert np.array_equal(bd.noise.numpy().astype(np.uint8), d['hardb'])
