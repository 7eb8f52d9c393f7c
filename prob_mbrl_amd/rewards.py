"""Analytic reward modules with the reference's constructor signatures and
parameter names (envs/cartpole/env.py:27-86, envs/pendulum/env.py:27-79,
envs/double_cartpole/env.py:27-91, envs/cart_acrobot/env.py:27-89,
envs/rendezvous/env.py:26-45, losses.py:67-75).

Each one restates its reference `forward` as the generic constants the fused
kernels evaluate (see include/pmbrl.h, pmbrl_reward):
    phi = expand ? [others, sin(angles), cos(angles)] : x
    delta = (C phi - C phi(target)) / norm
    r = exp(-w (delta' Q delta + u' R u))   |   -w (...)
"""
import numpy as np
import torch
from torch import nn


def _expand_np(x, angle_dims):
    x = np.asarray(x, dtype=np.float32)
    od = [i for i in range(x.shape[-1]) if i not in angle_dims]
    return np.concatenate([x[..., od], np.sin(x[..., angle_dims]), np.cos(x[..., angle_dims])], -1)


class AnalyticReward(nn.Module):
    """Base: subclasses fill angle_dims and `_tip_matrix()` ([k, De])."""
    kind = 'exp'
    weight = 0.5
    angle_dims = []

    def _frozen(self, value):
        value = value if isinstance(value, torch.Tensor) else torch.as_tensor(value)
        return nn.Parameter(value.detach().clone().float(), requires_grad=False)

    def _tip_matrix(self):
        raise NotImplementedError

    def _norm(self):
        return 1.0

    def _target_raw(self):
        return self.target.detach().cpu().numpy().reshape(-1)

    def spec(self, D):
        """Constants for states of width D: the raw width (angles expanded inside the
        reward) or the already-expanded width (envs/cartpole/env.py:60-63).  Memoised on the
        constants' storage and version counters: reading them from a device tensor is a
        host-device sync, and mc_pilco asks once per iteration."""
        sig = (D,) + tuple((p.data_ptr(), p._version, str(p.device)) for p in self.parameters())
        hit = getattr(self, '_spec_cache', None)
        if hit is not None and hit[0] == sig:
            return hit[1]
        out = self._spec_uncached(D)
        self._spec_cache = (sig, out)
        return out

    def _spec_uncached(self, D):
        C = np.asarray(self._tip_matrix(), dtype=np.float64)
        De = C.shape[1]
        na = len(self.angle_dims)
        if D == De:
            expand = False
        elif D == De - na:
            expand = True
        else:
            raise ValueError('%s: state width %d matches neither the raw (%d) nor the expanded '
                             '(%d) layout' % (type(self).__name__, D, De - na, De))
        ta = _expand_np(self._target_raw(), list(self.angle_dims)) if na else \
            np.asarray(self._target_raw(), dtype=np.float32)
        return dict(kind=self.kind, expand=expand, angle_dims=list(self.angle_dims), C=C,
                    tip_target=C @ ta.astype(np.float64), norm=float(self._norm()),
                    w=float(self.weight), Q=self.Q.detach().cpu().numpy().astype(np.float64),
                    R=self.R.detach().cpu().numpy().astype(np.float64))

    def forward(self, x, u):
        """Stand-alone evaluation (what env.reward_func(x, u) does in the reference, outside
        the rollout): a few small torch ops on whatever device x lives on.  Inside the fused
        rollout the same constants are evaluated by the kernels."""
        sp = self.spec(x.shape[-1])
        kw = dict(dtype=x.dtype, device=x.device)
        xa = x
        if sp['expand']:
            ad = list(sp['angle_dims'])
            od = [i for i in range(x.shape[-1]) if i not in ad]
            xa = torch.cat([x[..., od], x[..., ad].sin(), x[..., ad].cos()], -1)
        Cm = torch.as_tensor(sp['C'], **kw)
        delta = (xa @ Cm.t() - torch.as_tensor(sp['tip_target'], **kw)) / sp['norm']
        Q = torch.as_tensor(sp['Q'], **kw)
        R = torch.as_tensor(sp['R'], **kw)
        cost = sp['w'] * (((delta @ Q) * delta).sum(-1, keepdim=True) + ((u @ R) * u).sum(-1, keepdim=True))
        return (-cost).exp() if sp['kind'] == 'exp' else -cost


class CartpoleReward(AnalyticReward):
    angle_dims = [2]

    def __init__(self, pole_length=0.5, target=torch.tensor([0, 0, np.pi, 0]),
                 Q=16.0 * torch.eye(2), R=1e-4 * torch.eye(1)):
        super().__init__()
        self.Q, self.R = self._frozen(Q), self._frozen(R)
        self.target = self._frozen(target.unsqueeze(0) if target.dim() == 1 else target)
        self.pole_length = self._frozen(pole_length)

    def _tip_matrix(self):
        l = float(self.pole_length)
        C = np.zeros((2, 5))
        C[0, 0], C[0, 3], C[1, 4] = 1.0, l, -l      # [x + l sin(th), -l cos(th)]
        return C

    def _norm(self):
        return 2 * float(self.pole_length)


class PendulumReward(AnalyticReward):
    angle_dims = [0]

    def __init__(self, pole_length=1.0, target=torch.tensor([np.pi, 0]), Q=4.0 * torch.eye(2),
                 R=1e-4 * torch.eye(1)):
        super().__init__()
        self.Q, self.R = self._frozen(Q), self._frozen(R)
        self.target = self._frozen(target.unsqueeze(0) if target.dim() == 1 else target)
        self.pole_length = self._frozen(pole_length)

    def _tip_matrix(self):
        l = float(self.pole_length)
        C = np.zeros((2, 3))
        C[0, 1], C[1, 2] = l, -l                     # [l sin(th), -l cos(th)]
        return C

    def _norm(self):
        return 2 * float(self.pole_length)


class DoubleCartpoleReward(AnalyticReward):
    angle_dims = [2, 4]

    def __init__(self, pole1_length=0.6, pole2_length=0.6,
                 target=torch.tensor([0.0, 0.0, 0.0, 0.0, 0.0, 0.0]), Q=8.0 * torch.eye(2),
                 R=1e-3 * torch.eye(1)):
        super().__init__()
        self.Q, self.R = self._frozen(Q), self._frozen(R)
        self.target = self._frozen(target.unsqueeze(0) if target.dim() == 1 else target)
        self.pole1_length = self._frozen(pole1_length)
        self.pole2_length = self._frozen(pole2_length)

    def _tip_matrix(self):
        l1, l2 = float(self.pole1_length), float(self.pole2_length)
        C = np.zeros((2, 8))   # expanded: [x0, x1, x3, x5, sin2, sin4, cos2, cos4]
        C[0, 0], C[0, 4], C[0, 5] = 1.0, -l1, -l2
        C[1, 6], C[1, 7] = l1, l2
        return C

    def _norm(self):
        return 2 * (float(self.pole1_length) + float(self.pole2_length))


class CartAcrobotReward(DoubleCartpoleReward):
    """envs/cart_acrobot/env.py:27-89: same body as the double cartpole."""


class RendezvousReward(AnalyticReward):
    """envs/rendezvous/env.py:26-45: r = -(delta' Q delta + u' R u), delta = relative state."""
    kind = 'neg'
    weight = 1.0
    angle_dims = []

    def __init__(self, Q=1.0 * torch.eye(4), R=1.0 * torch.eye(4)):
        super().__init__()
        self.Q, self.R = self._frozen(Q), self._frozen(R)
        self.target = self._frozen(torch.zeros(1, 8))

    def _tip_matrix(self):
        C = np.zeros((4, 8))
        for i, (a, b) in enumerate([(0, 2), (1, 3), (4, 6), (5, 7)]):
            C[i, a], C[i, b] = 1.0, -1.0
        return C


class QuadraticSaturatingReward(AnalyticReward):
    """losses.py:67-75 as a reward: exp(-1/2 (x-target)' Q (x-target))  (= 1 - loss)."""
    angle_dims = []

    def __init__(self, target, Q, R=None):
        super().__init__()
        target = torch.as_tensor(target).float()
        self.target = self._frozen(target.unsqueeze(0) if target.dim() == 1 else target)
        self.Q = self._frozen(Q)
        self._R = R

    def bind_action_dim(self, U):
        self.R = self._frozen(torch.zeros(U, U) if self._R is None else self._R)

    def _tip_matrix(self):
        return np.eye(self.target.shape[-1])


class LinearFeatureReward(AnalyticReward):
    """Generic exp(-w ||C x - c||^2_Q - w u'Ru) (synthetic config 5)."""

    def __init__(self, C, c, Q, R, weight=0.5, kind='exp'):
        super().__init__()
        self.Cm = self._frozen(torch.as_tensor(C))
        self.c = self._frozen(torch.as_tensor(c))
        self.Q, self.R = self._frozen(Q), self._frozen(R)
        self.weight, self.kind = weight, kind

    def spec(self, D):
        C = self.Cm.detach().cpu().numpy().astype(np.float64)
        assert C.shape[1] == D
        return dict(kind=self.kind, expand=False, angle_dims=[], C=C,
                    tip_target=self.c.detach().cpu().numpy().astype(np.float64), norm=1.0,
                    w=float(self.weight), Q=self.Q.detach().cpu().numpy().astype(np.float64),
                    R=self.R.detach().cpu().numpy().astype(np.float64))


# ---------------------------------------------------------------------------
# reference-shaped reward modules
# ---------------------------------------------------------------------------
# class name of the reference module (envs/<env>/env.py) -> (class here, names of its length parameters)
_BY_NAME = {
    'CartpoleReward': (CartpoleReward, ('pole_length',)),
    'PendulumReward': (PendulumReward, ('pole_length',)),
    'DoubleCartpoleReward': (DoubleCartpoleReward, ('pole1_length', 'pole2_length')),
    'CartAcrobotReward': (CartAcrobotReward, ('pole1_length', 'pole2_length')),
    'RendezvousReward': (RendezvousReward, ()),
}


def from_module(obj):
    """The analytic reward of this build for a REFERENCE-shaped reward module -- what `env.reward_func` is in the
    reference's examples (examples/deep_pilco_mm.py:94-99,135): a torch module named after its environment holding
    the constants of its closed form as parameters (envs/cartpole/env.py:27-40: Q, R, target, pole_length; likewise
    pendulum, double_cartpole, cart_acrobot, rendezvous).  The constants are read from the module (not assumed), so
    non-default lengths / weights / targets carry over.  Returns `obj` itself if it already is one of ours, None if
    it is not a module this build recognises (the caller then treats it as a learned / unknown reward)."""
    if obj is None or isinstance(obj, AnalyticReward):
        return obj
    hit = _BY_NAME.get(type(obj).__name__)
    if hit is None or not all(hasattr(obj, a) for a in ('Q', 'R') + hit[1]):
        return None
    cls, lengths = hit
    kw = {a: getattr(obj, a).detach().clone() for a in lengths}
    kw.update(Q=obj.Q.detach().clone(), R=obj.R.detach().clone())
    if hasattr(obj, 'target') and cls is not RendezvousReward:
        kw['target'] = obj.target.detach().clone()
    out = cls(**kw)
    return out.to(obj.Q.device)
