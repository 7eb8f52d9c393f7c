"""Host data path of the policy-search loop: the experience container, the
priority sum-tree and the real-system execution loop, with the reference's call surface and
on-disk format (utils/experience_dataset.py:9-367, utils/apply_controller.py:6-94,
utils/core.py:200-226), so that `experience.pth.tar` / `latest_*.pth.tar` written by either
code base load in the other.  Plain numpy / torch host code -- nothing here is on the device
hot path."""
import os
import time
import warnings
from collections.abc import Iterable

import numpy as np
import torch

from .utils import to_complex


class ExperienceDataset(torch.nn.Module):
    """Episodes of (state, action, reward, done, info, time stamp) as nested Python lists
    (utils/experience_dataset.py:9-268)."""

    _FIELDS = ('states', 'actions', 'rewards', 'info', 'done', 'time_stamps', 'curr_episode',
               'policy_parameters')

    def __init__(self, name='Experience'):
        super().__init__()
        self.name = name
        self._clear()
        self.done = []
        self.state_changed = True

    def _clear(self):
        self.time_stamps, self.states, self.actions, self.rewards = [], [], [], []
        self.info, self.policy_parameters = [], []
        self.curr_episode = -1

    # -- building ------------------------------------------------------------------
    def new_episode(self, policy_params=None):
        # (the reference forgets `done` here, so its add_sample raises on a fresh dataset)
        for lst in (self.time_stamps, self.states, self.actions, self.rewards, self.info, self.done):
            lst.append([])
        self.policy_parameters.append(policy_params if policy_params else [])
        self.curr_episode += 1
        self.state_changed = True

    def add_sample(self, x_t=None, u_t=None, c_t=None, done=None, info=None, t=None):
        if self.curr_episode < 0:
            self.new_episode()
        e = self.curr_episode
        self.states[e].append(x_t)
        self.actions[e].append(u_t)
        self.rewards[e].append(c_t)
        self.done[e].append(done)
        self.info[e].append(info)
        self.time_stamps[e].append(t)
        self.state_changed = True

    def append_episode(self, states, actions, rewards, dones=None, infos=None, policy_params=None, ts=None):
        for lst, val in ((self.policy_parameters, policy_params), (self.done, dones), (self.info, infos),
                         (self.time_stamps, ts)):
            if val is not None:
                lst.append(val)
        self.states.append(states)
        self.actions.append(actions)
        self.rewards.append(rewards)
        self.curr_episode += 1

    def n_samples(self):
        return sum(len(ep) for ep in self.states)

    def n_episodes(self):
        return len(self.states)

    def reset(self):
        self._clear()
        self.state_changed = False

    def truncate(self, episode):
        if 0 < episode <= self.curr_episode:
            self.curr_episode = episode
            for f in ('time_stamps', 'states', 'actions', 'rewards', 'done', 'info', 'policy_parameters'):
                setattr(self, f, getattr(self, f)[episode:])

    # -- views ---------------------------------------------------------------------
    def get_dynmodel_dataset(self, deltas=True, filter_episodes=None, angle_dims=None, x_steps=1,
                             u_steps=1, output_steps=1, return_costs=False, stack=False):
        """(inputs, targets) for the dynamics model (utils/experience_dataset.py:122-234).
        Row r of an episode: inputs = the last x_steps states (angle dims expanded, the first state
        repeated before the episode start) and the last u_steps actions (zeros before the start),
        oldest first; targets = the next output_steps state changes (or states), then, if asked,
        the rewards of those steps.  stack=True keeps the step axis instead of concatenating."""
        angle_dims = list(angle_dims) if angle_dims is not None else []
        if filter_episodes is None:
            filter_episodes = []
        if not isinstance(filter_episodes, list):
            filter_episodes = [filter_episodes]
        episodes = filter_episodes if len(filter_episodes) else list(range(self.n_episodes()))
        if stack:
            u_steps = x_steps
            output_steps = x_steps + output_steps - 1
        X, Y = [], []
        for e in episodes:
            if len(self.states[e]) == 0:
                continue
            S = torch.as_tensor(np.asarray(self.states[e])).double()
            A = torch.as_tensor(np.asarray(self.actions[e])).double()
            T = S.shape[0]
            n = T - output_steps
            Sx = to_complex(S, angle_dims)
            rows = torch.arange(n)

            def lagged(M, steps, pad_first):
                cols = []
                for k in range(steps):                      # oldest first
                    t = rows - (steps - 1) + k
                    blk = M[t.clamp(min=0)]
                    if not pad_first:
                        blk = blk * (t >= 0).to(M.dtype).unsqueeze(-1)
                    cols.append(blk)
                return torch.stack(cols, 1) if stack else torch.cat(cols, 1)

            inp = torch.cat([lagged(Sx, x_steps, True), lagged(A, u_steps, False)], -1)
            fut = [S[rows + 1 + k] - S[rows + k] if deltas else S[rows + 1 + k] for k in range(output_steps)]
            tgt = torch.stack(fut, 1) if stack else torch.cat(fut, 1)
            if return_costs:
                R = torch.as_tensor(np.asarray(self.rewards[e])).double().squeeze(-1)
                if R.dim() == 1:
                    R = R.unsqueeze(1)
                rc = [R[rows + k] for k in range(output_steps)]
                tgt = torch.cat([tgt, torch.stack(rc, 1) if stack else torch.cat(rc, 1)], -1)
            X.append(inp)
            Y.append(tgt)
        return torch.cat(X).detach(), torch.cat(Y).detach()

    def sample_states(self, n_samples=1, timestep=0):
        """n_samples states drawn uniformly (np.random) from the given time step(s) of every
        episode, or from all stored states (timestep=None); float64 like the reference."""
        if timestep is None:
            pool = np.concatenate([np.asarray(ep) for ep in self.states])
        else:
            steps = timestep if isinstance(timestep, Iterable) else [timestep]
            pool = np.concatenate([[ep[t] for t in steps if t < len(ep)] for ep in self.states])
        idx = np.random.choice(range(len(pool)), n_samples)
        return torch.tensor(pool)[idx].double()

    # -- persistence (same keys as utils/experience_dataset.py:251-268) ---------------
    def save(self, filename):
        d = os.path.dirname(filename)
        if d:
            os.makedirs(d, exist_ok=True)
        torch.save({k: getattr(self, k) for k in self._FIELDS}, filename)

    def load(self, filename):
        try:
            sd = torch.load(filename, weights_only=False)
        except TypeError:
            sd = torch.load(filename)
        self.__dict__.update(sd)


class SumTree:
    """Binary sum tree over max_size priorities (utils/experience_dataset.py:271-367): leaf i of
    the data array sits at tree index i + max_size - 1; `sample` draws one leaf per equal-mass
    segment and returns importance weights (N p)^-beta / max."""

    def __init__(self, max_size):
        self.max_size = max_size
        self.data = [None] * max_size
        self.sum_tree = np.zeros(2 * max_size - 1)
        self.counts = np.zeros(max_size)
        self.idx = 0
        self.max_p = 1.0
        self.max_count = 0
        self.size = 0
        self.norm_factor = 1.0

    def append(self, data, priority):
        slot = self.idx
        self.data[slot] = data
        self.counts[slot] = 1
        self.update(slot + self.max_size - 1, priority)
        self.idx = (slot + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def update(self, idx, priority):
        self.sum_tree[idx] = priority * self.norm_factor
        node = idx
        while node > 0:                       # refresh the ancestors
            node = (node - 1) // 2
            self.sum_tree[node] = self.sum_tree[2 * node + 1] + self.sum_tree[2 * node + 2]
        self.max_p = max(self.max_p, priority)

    def renormalize(self):
        scale = 1.0 / self.sum_tree[0]
        self.norm_factor *= scale
        self.sum_tree *= scale

    def _descend(self, mass):
        """Leaf reached by walking down with cumulative mass `mass` (scalar or array)."""
        mass = np.array(mass, dtype=np.float64, copy=True)
        scalar = mass.ndim == 0
        mass = np.atleast_1d(mass)
        node = np.zeros(len(mass), dtype=np.int64)
        n_nodes = len(self.sum_tree)
        while True:
            left = 2 * node + 1
            live = left < n_nodes
            if not live.any():
                break
            lv = self.sum_tree[np.where(live, left, 0)]
            go_left = mass <= lv
            nxt = np.where(go_left, left, left + 1)
            mass = np.where(live & ~go_left, mass - lv, mass)
            node = np.where(live, nxt, node)
        return int(node[0]) if scalar else node

    def get(self, priority):
        idx = self._descend(priority)
        return [idx, self.sum_tree[idx], self.data[idx - self.max_size + 1]]

    def get_batch(self, priority):
        idxs = self._descend(np.atleast_1d(priority))
        return idxs, self.sum_tree[idxs], [self.data[i] for i in idxs - self.max_size + 1]

    def sample(self, batchsize, beta=1.0):
        total = self.sum_tree[0]
        seg = total / batchsize
        mass = (np.arange(batchsize) + np.random.rand(batchsize)) * seg
        idxs, pri, samples = self.get_batch(mass)
        pri = np.asarray(pri)
        leaf = idxs - self.max_size + 1
        self.counts[leaf] += 1
        self.max_count = max(self.max_count, self.counts[leaf].max())
        w = (self.size * (pri / total))**-beta
        return samples, idxs, w / w.max()

    def replay_sample(self, idxs):
        """What `sample` leaves behind and hands out when the draw (tree indices idxs) was made elsewhere -- by rank 0
        of a sharded run, whose replicas of the tree must see the same draw whatever each rank's numpy generator did
        in between: the visit counts and the stored start states of those leaves."""
        idxs = np.asarray(idxs, dtype=np.int64)
        leaf = idxs - self.max_size + 1
        self.counts[leaf] += 1
        self.max_count = max(self.max_count, self.counts[leaf].max())
        return [self.data[i] for i in leaf]


def apply_controller(env, policy, max_steps, preprocess=None, callback=None, realtime=False,
                     stop_when_done=True):
    """Run `policy` on `env` for up to max_steps steps and return the trajectory
    (utils/apply_controller.py:6-94): (states, actions, costs, dones, infos)."""
    tag = 'apply_controller'
    if hasattr(policy, 'get_params'):
        if len(policy.get_params()) == 0:
            policy.init_params()
        policy(np.zeros((policy.D, )))
    print(tag, 'Starting run')
    dt = getattr(env, 'dt', None)
    if dt is not None:
        print(tag, 'Running for %f seconds' % (max_steps * dt))
    else:
        print(tag, 'Running for %d steps' % max_steps)
    x_t = env.reset()
    traj = []
    tick = t_start = time.time()
    t = -1
    for t in range(max_steps):
        obs = preprocess(x_t) if callable(preprocess) else x_t
        u_t = policy(obs, t=t)
        u_t = (u_t[0] if isinstance(u_t, (list, tuple)) else u_t).flatten()
        x_next, c_t, done, info = env.step(u_t)
        info['done'] = done
        info['t'] = t * dt if realtime else tick - t_start
        traj.append((x_t, u_t, c_t, done, info))
        if callable(callback):
            callback(x_t, u_t, c_t, done, info)
        if done and stop_when_done:
            break
        x_t = x_next
        if realtime:
            time.sleep(max(float(dt - (time.time() - tick)), 0))
        tick = time.time()
    states, actions, costs, dones, infos = zip(*traj)
    msg = 'Done after [%d] steps. Stopping robot.' % (t + 1)
    if all(c is not None for c in costs):
        msg += ' Value of run [%f]' % np.array(costs).sum()
    print(tag, msg)
    if hasattr(env, 'stop'):
        env.stop()
    return states, actions, costs, dones, infos


def load_checkpoint(path, dyn, pol, exp, val=None):
    """utils/core.py:200-226: best-effort restore of latest_{dynamics,policy,critic}.pth.tar and
    experience.pth.tar from `path` (a missing or unreadable file only warns)."""
    def _load(fname, into):
        full = os.path.join(path, fname)
        try:
            try:
                sd = torch.load(full, weights_only=False)
            except TypeError:
                sd = torch.load(full)
            into.load(sd)
        except Exception:
            warnings.warn('Unable to load parameters at {}'.format(full))

    _load('latest_dynamics.pth.tar', dyn)
    _load('latest_policy.pth.tar', pol)
    if val is not None:
        _load('latest_critic.pth.tar', val)
    try:
        exp.load(os.path.join(path, 'experience.pth.tar'))
    except Exception:
        warnings.warn('Unable to load experience at {}'.format(os.path.join(path, 'experience.pth.tar')))
