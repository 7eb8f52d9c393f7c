"""A self-contained cart-pole swing-up task for the examples and the end-to-end test.

The reference's environments are gym / Box2D wrappers around ODE simulators
(prob_mbrl/envs, SURVEY.md section 2: out of scope, not installed here).  This is NOT one of
them: it is a small numpy simulator written from the textbook equations of motion of a cart
with a uniform-rod pendulum, exposing the handful of attributes the example script and
`apply_controller` use (reset / step / seed, observation_space / action_space with shape, low,
high, sample(), reward_func, dt, spec).  State = [x, dx, theta, dtheta], theta = 0 hanging down;
`rewards.CartpoleReward` (target theta = pi) is the task's reward.
"""
import numpy as np

from . import rewards


class _Box:
    def __init__(self, low, high, rng):
        self.low = np.asarray(low, dtype=np.float32)
        self.high = np.asarray(high, dtype=np.float32)
        self.shape = self.low.shape
        self._rng = rng

    def sample(self):
        return self._rng.uniform(self.low, self.high).astype(np.float32)


class Cartpole:
    """Cart (mass M) on a track with friction b, uniform rod (mass m, length l) on a free pivot."""
    spec = None

    def __init__(self, pole_length=0.5, pole_mass=0.5, cart_mass=0.5, friction=0.1, gravity=9.82,
                 dt=0.1, substeps=10, max_force=10.0, init_std=(0.1, 0.1, 0.1, 0.1)):
        self.l, self.m, self.M, self.b, self.g = pole_length, pole_mass, cart_mass, friction, gravity
        self.dt, self.substeps = dt, substeps
        self._rng = np.random.RandomState(0)
        hi = np.array([np.inf] * 4)
        self.observation_space = _Box(-hi, hi, self._rng)
        self.action_space = _Box([-max_force], [max_force], self._rng)
        self.init_std = np.asarray(init_std, dtype=np.float64)
        self.reward_func = rewards.CartpoleReward(pole_length=pole_length)
        self.state = np.zeros(4)

    def seed(self, seed=None):
        self._rng.seed(seed)
        return [seed]

    def _deriv(self, s, u):
        x, v, th, w = s
        l, m, M, b, g = self.l, self.m, self.M, self.b, self.g
        sn, cs = np.sin(th), np.cos(th)
        den = 4.0 * (M + m) - 3.0 * m * cs * cs
        dv = (2.0 * m * l * w * w * sn + 3.0 * m * g * sn * cs + 4.0 * u - 4.0 * b * v) / den
        dw = (-3.0 * m * l * w * w * sn * cs - 6.0 * (M + m) * g * sn - 6.0 * (u - b * v) * cs) / (l * den)
        return np.array([v, dv, w, dw])

    def reset(self):
        self.state = self.init_std * self._rng.randn(4)
        return self.state.astype(np.float32)

    def step(self, action):
        u = float(np.clip(np.asarray(action).reshape(-1)[0], self.action_space.low[0],
                          self.action_space.high[0]))
        h = self.dt / self.substeps
        s = self.state
        for _ in range(self.substeps):          # classical Runge-Kutta, zero-order hold on u
            k1 = self._deriv(s, u)
            k2 = self._deriv(s + 0.5 * h * k1, u)
            k3 = self._deriv(s + 0.5 * h * k2, u)
            k4 = self._deriv(s + h * k3, u)
            s = s + (h / 6.0) * (k1 + 2 * k2 + 2 * k3 + k4)
        self.state = s
        obs = s.astype(np.float32)
        import torch
        r = self.reward_func(torch.tensor(obs)[None], torch.tensor([[u]], dtype=torch.float32))
        return obs, float(r.reshape(-1)[0]), False, {}
