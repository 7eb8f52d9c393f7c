"""ORACLE (test infrastructure, NOT product code).

Autograd-free numpy statement of the rollout forward pass and its explicit
adjoint, organised exactly like the HIP kernels:

  forward  : per step, stash  actT (policy layer inputs), active bits,
             T_pol = z_pi*e*sigmoid(-l+c), T_dyn = z_f*e'*sigmoid(-l'+c),
             pre-moment-matching x~ and r~
  backward : reverse sweep producing the per-(t,row) pre-activation gradients
             G_l of every policy layer (the "G stash"), then
             dW_l = sum_{t,b} G_l^T act_l,  db_l = sum G_l   (the dW GEMM).

It exists so that (a) the formulas the kernels implement are pinned against the
reference's autograd result (tests/test_oracle_golden.py compares it with the
golden fixtures generated from /root/reference) and (b) kernel intermediates
can be compared one by one when debugging on the GPU.

Reference lines restated: utils/rollout.py:20-29,62-163; models/core.py:169-187,
221-248,265-303; models/modules.py:46-61,120-160; models/densities.py:87-121;
envs/cartpole/env.py:41-86; algorithms/mc_pilco.py:134-144,190-197.
"""
import numpy as np

LOG_MAX_STD = np.log(5.0)


def _mm(a, b, role='fwd'):
    """Every dense product of the path goes through here (role: 'fwd' layer, 'dx' adjoint chain,
    'dw' weight gradient) so that tools/split_precision_study.py can swap in an emulation of the
    split-bf16 matrix-core arithmetic; the oracle itself is the plain product."""
    return a @ b


def softplus(x):
    return np.where(x > 20, x, np.log1p(np.exp(np.minimum(x, 20))))


def sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def expand_angles(x, adims):
    adims = list(adims)
    odims = [i for i in range(x.shape[-1]) if i not in adims]
    return np.concatenate([x[..., odims], np.sin(x[..., adims]),
                           np.cos(x[..., adims])], -1), odims


class Problem:
    """Plain-numpy view of a fixture dict (see tools/make_golden.py)."""

    def __init__(self, d, dtype=np.float64):
        f = lambda k: np.asarray(d[k], dtype=dtype)  # noqa: E731
        self.dtype = dtype
        npl, ndl = int(d['pol_n_layers']), int(d['dyn_n_layers'])
        self.pW = [f('pol_W%d' % i) for i in range(npl)]
        self.pb = [f('pol_b%d' % i) for i in range(npl)]
        self.pmask = [f('pol_mask%d' % i) for i in range(npl - 1)]
        self.pkeep = [dtype(np.float32(k)) for k in np.asarray(d['pol_keep'])]
        self.pz = f('pol_z')
        self.pscale, self.pbias = f('pol_scale'), f('pol_bias')
        self.dW = [f('dyn_W%d' % i) for i in range(ndl)]
        self.db = [f('dyn_b%d' % i) for i in range(ndl)]
        self.dmask = [f('dyn_mask%d' % i) for i in range(ndl - 1)]
        self.dkeep = [dtype(np.float32(k)) for k in np.asarray(d['dyn_keep'])]
        self.dz = f('dyn_z')
        self.mx, self.iSx, self.my, self.Sy = (f('dyn_mx'), f('dyn_iSx'),
                                               f('dyn_my'), f('dyn_Sy'))
        assert 'dyn_gmm_n' not in d or int(d['dyn_gmm_n']) <= 1, 'mixture head: torch oracle only'
        # angle_dims of Policy / the dynamics Regressor (models/core.py:233-234,173-174)
        self.pol_adims = [int(a) for a in np.asarray(d['pol_angle_dims'])]
        self.dyn_adims = [int(a) for a in np.asarray(d['dyn_angle_dims'])]
        self.rew_kind = str(d['rew_kind'])
        self.rew_expand = bool(d['rew_expand'])
        self.rew_adims = [int(a) for a in np.asarray(d['rew_angle_dims'])]
        self.C, self.tt = f('rew_C'), f('rew_tip_target')
        self.norm, self.w = dtype(d['rew_norm']), dtype(d['rew_w'])
        self.Q, self.R = f('rew_Q'), f('rew_R')
        self.x0 = f('x0')
        self.H = int(d['H'])
        self.gamma = f('gamma')
        self.mm_states, self.mm_rewards = bool(d['mm_states']), bool(d['mm_rewards'])
        G = int(d['mm_groups'])
        self.G = G if G > 0 else 1
        self.maximize = bool(d['maximize'])
        self.infer_ns = bool(d['infer_ns']) if 'infer_ns' in d else False
        self.z_mm = f('z_mm') if 'z_mm' in d else None
        self.z_rr = f('z_rr') if 'z_rr' in d else None
        # shard placement in a larger global batch (multi-GPU decomposition)
        self.row_offset = 0
        self.Bg = self.x0.shape[0]

    def shard(self, lo, hi, G_local):
        """View of rows [lo, hi) of this problem as one rank's shard."""
        import copy
        q = copy.copy(self)
        q.x0 = self.x0[lo:hi]
        q.pmask = [m[lo:hi] for m in self.pmask]
        q.dmask = [m[lo:hi] for m in self.dmask]
        q.pz, q.dz = self.pz[lo:hi], self.dz[lo:hi]
        q.row_offset, q.Bg = lo, self.x0.shape[0]
        q.G = G_local if G_local else 1
        return q


# ---------------------------------------------------------------------------
# moment matching (utils/rollout.py:20-29) and its adjoint
# ---------------------------------------------------------------------------
def expand_bwd(g, x, adims):
    """Adjoint of expand_angles: g over [others | sin | cos] -> gradient over x."""
    adims = list(adims)
    if not adims:
        return g
    od = [i for i in range(x.shape[-1]) if i not in adims]
    no, na = len(od), len(adims)
    gx = np.zeros_like(x)
    gx[:, od] = g[:, :no]
    gx[:, adims] = g[:, no:no + na] * np.cos(x[:, adims]) - g[:, no + na:] * np.sin(x[:, adims])
    return gx


def mm_forward(s, z, infer_ns=False):
    """s, z: [M, d].  Returns out and the cache (delta, L, zhat)."""
    M, d = s.shape
    m = s.mean(0, keepdims=True)
    delta = s - m
    S = delta.T @ delta / (M - 1) + 1e-12 * np.eye(d)
    L = np.linalg.cholesky(S)
    if infer_ns:
        zhat = np.linalg.solve(L, delta.T).T
    else:
        zhat = (z - z.mean(0, keepdims=True)) / z.std(0, ddof=1, keepdims=True)
    return m + zhat @ L.T, (delta, L, zhat)


def mm_backward(g, cache):
    """Adjoint of mm_forward w.r.t. s (zhat constant).  SURVEY Appendix A."""
    delta, L, zhat = cache
    M, d = delta.shape
    mbar = g.sum(0, keepdims=True)
    Lbar = np.tril(g.T @ zhat)
    P = np.tril(L.T @ Lbar)
    P[np.diag_indices(d)] *= 0.5
    Linv = np.linalg.inv(L)
    Sbar = Linv.T @ P @ Linv
    Sbar = 0.5 * (Sbar + Sbar.T)
    dbar = delta @ (2.0 * Sbar) / (M - 1)
    return dbar - dbar.mean(0, keepdims=True) + mbar / M


# ---------------------------------------------------------------------------
# forward with stashes
# ---------------------------------------------------------------------------
def mlp_fwd(x, Ws, bs, masks, keeps):
    """Returns output, list of layer inputs, list of active bits (mask & pre>0)."""
    acts, bits = [x], []
    h = x
    n = len(Ws)
    for i in range(n):
        p = _mm(h, Ws[i].T) + bs[i]
        if i < n - 1:
            act = (p > 0) & (masks[i][:h.shape[0]] > 0)
            h = np.where(act, p, 0.0)
            if keeps[i] != 1.0:
                h = h / keeps[i]
            bits.append(act)
            acts.append(h)
        else:
            h = p
    return h, acts, bits


def reward_fwd(P, xt, a):
    if P.rew_expand:
        phi, odims = expand_angles(xt, P.rew_adims)
    else:
        phi, odims = xt, None
    delta = (phi @ P.C.T - P.tt) / P.norm
    cost = P.w * (np.sum((delta @ P.Q) * delta, -1, keepdims=True) +
                  np.sum((a @ P.R) * a, -1, keepdims=True))
    r = np.exp(-cost) if P.rew_kind == 'exp' else -cost
    return r, delta


def reward_bwd(P, xt, a, r, delta, gr):
    """Returns (g wrt x~, g wrt a) of the reward term."""
    gc = -gr * r if P.rew_kind == 'exp' else -gr
    gdelta = gc * P.w * (delta @ (P.Q + P.Q.T))
    ga = gc * P.w * (a @ (P.R + P.R.T))
    gphi = (gdelta / P.norm) @ P.C
    if not P.rew_expand:
        return gphi, ga
    ad = P.rew_adims
    od = [i for i in range(xt.shape[-1]) if i not in ad]
    no, na = len(od), len(ad)
    gx = np.zeros_like(xt)
    gx[:, od] = gphi[:, :no]
    gx[:, ad] = gphi[:, no:no + na] * np.cos(xt[:, ad]) - \
        gphi[:, no + na:] * np.sin(xt[:, ad])
    return gx, ga


def forward(P):
    B, D = P.x0.shape
    H, G = P.H, P.G
    M = B // G
    x = P.x0.copy()
    st = dict(states=[x], actions=[], rewards=[], pacts=[], pbits=[], dbits=[],
              Tp=[], Td=[], xt=[], rt=[], mmc_s=[], mmc_r=[])
    for t in range(H):
        # [H, B, h] masks: a fresh draw per step (resample_policy / resample_model=True)
        pmask = [m[t] if m.ndim == 3 else m for m in P.pmask]
        dmask = [m[t] if m.ndim == 3 else m for m in P.dmask]
        xp = expand_angles(x, P.pol_adims)[0] if P.pol_adims else x
        o, pacts, pbits = mlp_fwd(xp, P.pW, P.pb, pmask, P.pkeep)
        U = o.shape[1] // 2
        mu, l = o[:, :U], o[:, U:]
        lc = -softplus(-l + LOG_MAX_STD) + LOG_MAX_STD
        e = np.exp(lc)
        u = mu + P.pz[:B] * e
        th = np.tanh(u)
        a = P.pscale * th + P.pbias
        Tp = P.pz[:B] * e * sigmoid(-l + LOG_MAX_STD)
        xa = np.concatenate([x, a], 1)
        if P.dyn_adims:
            xa = expand_angles(xa, P.dyn_adims)[0]
        xin = (xa - P.mx) * P.iSx
        o2, _, dbits = mlp_fwd(xin, P.dW, P.db, dmask, P.dkeep)
        mu2, l2 = o2[:, :D], o2[:, D:]
        lc2 = -softplus(-l2 + LOG_MAX_STD) + LOG_MAX_STD + np.log(P.Sy)
        e2 = np.exp(lc2)
        xt = x + (mu2 * P.Sy + P.my + P.dz[:B] * e2)
        Td = P.dz[:B] * e2 * sigmoid(-l2 + LOG_MAX_STD)
        rt, _ = reward_fwd(P, xt, a)
        idx = (t + P.row_offset + np.arange(B)) % P.Bg
        xn, r = xt, rt
        mmc_s = mmc_r = None
        if P.mm_states:
            z1 = P.z_mm[idx]
            xn = np.empty_like(xt)
            mmc_s = []
            for g in range(G):
                sl = slice(g * M, (g + 1) * M)
                xn[sl], c = mm_forward(xt[sl], z1[sl], P.infer_ns)
                mmc_s.append(c)
        if P.mm_rewards:
            z2 = P.z_rr[idx]
            r = np.empty_like(rt)
            mmc_r = []
            for g in range(G):
                sl = slice(g * M, (g + 1) * M)
                r[sl], c = mm_forward(rt[sl], z2[sl], P.infer_ns)
                mmc_r.append(c)
        st['actions'].append(a)
        st['rewards'].append(r)
        st['pacts'].append(pacts)
        st['pbits'].append(pbits)
        st['dbits'].append(dbits)
        st['Tp'].append(Tp)
        st['Td'].append(Td)
        st['xt'].append(xt)
        st['rt'].append(rt)
        st['mmc_s'].append(mmc_s)
        st['mmc_r'].append(mmc_r)
        x = xn
        st['states'].append(x)
    return st


def loss_weights(P, B):
    """g[t,b] = dL/dr[t,b]  (algorithms/mc_pilco.py:134-144,190)."""
    sign = -1.0 if P.maximize else 1.0
    return sign * P.gamma[:, None] * np.ones((1, B)) / P.Bg


def backward(P, st, gr_all=None):
    """Returns (flat policy grad in torch parameter order, dL/dx0, G stash)."""
    B, D = P.x0.shape
    H, G = P.H, P.G
    M = B // G
    npl, ndl = len(P.pW), len(P.dW)
    if gr_all is None:
        gr_all = loss_weights(P, B)
    gW = [np.zeros_like(w) for w in P.pW]
    gb = [np.zeros_like(b) for b in P.pb]
    gx = np.zeros((B, D), dtype=P.dtype)
    Gst = []
    for t in reversed(range(H)):
        x, a = st['states'][t], st['actions'][t]
        xt, rt = st['xt'][t], st['rt'][t]
        gr = gr_all[t][:, None].astype(P.dtype)
        if P.mm_rewards:
            g2 = np.empty_like(gr)
            for g in range(G):
                sl = slice(g * M, (g + 1) * M)
                g2[sl] = mm_backward(gr[sl], st['mmc_r'][t][g])
            gr = g2
        gxt = gx
        if P.mm_states:
            g2 = np.empty_like(gx)
            for g in range(G):
                sl = slice(g * M, (g + 1) * M)
                g2[sl] = mm_backward(gx[sl], st['mmc_s'][t][g])
            gxt = g2
        _, delta = reward_fwd(P, xt, a)
        gx_r, ga = reward_bwd(P, xt, a, rt, delta, gr)
        gxt = gxt + gx_r
        # dynamics head and trunk (dX only -- dynamics weights are frozen)
        go = np.concatenate([gxt * P.Sy, gxt * st['Td'][t]], 1)
        gh = _mm(go, P.dW[ndl - 1], 'dx')
        for i in reversed(range(ndl - 1)):
            gp = np.where(st['dbits'][t][i], gh, 0.0)
            if P.dkeep[i] != 1.0:
                gp = gp / P.dkeep[i]
            gh = _mm(gp, P.dW[i], 'dx')
        gxa = expand_bwd(gh * P.iSx, np.concatenate([x, a], 1), P.dyn_adims)
        gx = gxt + gxa[:, :D]
        ga = ga + gxa[:, D:]
        # policy head
        th = (a - P.pbias) / P.pscale
        gu = ga * P.pscale * (1.0 - th * th)
        go = np.concatenate([gu, gu * st['Tp'][t]], 1)
        Gs = [None] * npl
        Gs[npl - 1] = go
        gW[npl - 1] += _mm(go.T, st['pacts'][t][npl - 1], 'dw')
        gb[npl - 1] += go.sum(0)
        gh = _mm(go, P.pW[npl - 1], 'dx')
        for i in reversed(range(npl - 1)):
            gp = np.where(st['pbits'][t][i], gh, 0.0)
            if P.pkeep[i] != 1.0:
                gp = gp / P.pkeep[i]
            Gs[i] = gp
            gW[i] += _mm(gp.T, st['pacts'][t][i], 'dw')
            gb[i] += gp.sum(0)
            gh = _mm(gp, P.pW[i], 'dx')
        gx = gx + expand_bwd(gh, x, P.pol_adims)
        Gst.append(Gs)
    flat = np.concatenate([np.concatenate([w.reshape(-1), b.reshape(-1)])
                           for w, b in zip(gW, gb)])
    return flat, gx, Gst[::-1]


def loss(P, st):
    B = P.x0.shape[0]
    g = loss_weights(P, B)
    return float(sum((g[t] * st['rewards'][t][:, 0]).sum() for t in range(P.H)))
