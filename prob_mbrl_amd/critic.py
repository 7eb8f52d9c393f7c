"""Value-function (critic) fit used as `on_rollout` hook with `mc_pilco(value_func=V)`:
`update_value_function` with the signature and semantics of the function the reference keeps in
its example script (examples/deep_pilco_no_mm_with_value.py:14-66), on the device:

    V0      = V(states[0])   in training mode (concrete dropout with straight-through samples)
    targets = sum_{j<H} discount(j) r_j + discount(H) V_tgt(states[H])          (constants)
    loss    = mse(V0, targets) + reg_weight * V.regularization_loss()

Forward + backward of the fit are ONE pmbrl_bnn_loss_grad call (loss_kind = MSE) on the
normalised inputs / targets, the optimiser step is pmbrl_clip_adam; V_tgt(states[H]) is a
stand-alone forward (pmbrl_mlp_forward).  Offered for a critic WITHOUT an output density and one
output (what the example builds); a density critic is a NotImplementedError.
"""
import numpy as np
import torch

from . import engine as E
from .algorithms import _adam_flat_state, _sync_adam_state
from .models import CDropout
from .train_regressor import flat_module_parameters

_STEPS = {}


def update_value_function(V, opt, H, i, states, actions, rewards, discount, V_target=None, reg_weight=1e-4,
                          resample=False, polyak_averaging=0.005, _bernoulli=None):
    """`_bernoulli` (tests): the {0,1} outcomes of the concrete-dropout draws, in the order the
    reference makes them (V0's layers, then V_tgt's), instead of fresh ones."""
    V_tgt = V if V_target is None else V_target
    V.train()
    dev = V.mx.device
    if dev.type != 'cuda':
        raise RuntimeError('the critic must live on a HIP device (no CPU fallback)')
    if V.output_density is not None:
        raise NotImplementedError('critic with an output density: not offered on the device path')
    linears, drops, inner = V.model.layer_spec()
    if inner is not None or linears[-1].out_features != 1:
        raise NotImplementedError('the device path fits a critic with one plain output')
    draws = list(_bernoulli) if _bernoulli is not None else None
    n_drop = sum(1 for d in drops if isinstance(d, CDropout))
    returns = torch.stack([r * discount(j) for j, r in enumerate(rewards[:H])]).sum(0).detach()
    x0 = states[0].detach().to(torch.float32)
    B = x0.shape[0]

    # ---- targets: V_tgt(states[H]) with its own draws (training mode if V_tgt is V, like the reference).
    #      First: a stand-alone forward re-points the Linear parameters at ITS flat buffer, the fit below
    #      then gathers all parameters (with the dropout logits) into its own.
    if draws is not None:
        for k, dr in enumerate([d for d in V_tgt.model._modules.values() if isinstance(d, CDropout)]):
            dr._forced_sample = torch.as_tensor(draws[n_drop + k], dtype=torch.float32, device=dev)
    try:
        VH = V_tgt(states[H].detach(), resample=resample)
    finally:
        for dr in V_tgt.model._modules.values():
            if hasattr(dr, '_forced_sample'):
                del dr._forced_sample
    targets = (returns + discount(H) * VH.detach()).to(torch.float32).reshape(B, 1)

    # ---- parameters in module order, the last layer padded with a zero log-std row for the kernel
    params, temps, rscale, dreg = [], [], [], []
    for l, lin in enumerate(linears):
        params += [lin.weight, lin.bias]
        if l < len(linears) - 1:
            dr = drops[l]
            if dr is None:
                temps.append(0.0); rscale.append(0.0); dreg.append(0.0)
            elif isinstance(dr, CDropout):
                if dr.logit_p.numel() != lin.out_features:
                    dr.logit_p.data = dr.logit_p.data.reshape(-1).expand(lin.out_features).clone()
                params.append(dr.logit_p)
                temps.append(float(dr.temp)); rscale.append(float(dr.regularizer_scale))
                dreg.append(float(dr.dropout_regularizer))
            else:
                raise NotImplementedError('the critic fit is offered for concrete dropout (CDropout) layers')
    flat = flat_module_parameters(params, V)
    cache = _adam_flat_state(opt, params, flat)
    if cache is None:
        raise NotImplementedError('the critic fit on the device needs a plain torch.optim.Adam over V.parameters()')
    K = linears[-1].in_features
    n_last = K + 1
    padded = torch.cat([flat[:-n_last], flat[-n_last:-1], flat.new_zeros(K), flat[-1:], flat.new_zeros(1)])
    dims = [linears[0].in_features] + [l.out_features for l in linears[:-1]] + [2]

    # ---- V0's noise: the stored uniform noise (redrawn if it cannot be reused), fresh Bernoulli draws
    us, bvars = [], []
    for l, dr in enumerate(drops):
        if not isinstance(dr, CDropout):
            continue
        h = linears[l].out_features
        n = dr.noise
        if resample:
            n = torch.rand(B, h, device=dev)
        elif n.dim() != 2 or n.shape[1] != h or n.shape[0] < B:
            dr.update_noise(torch.empty(B, h))
            n = dr.noise
        us.append(n[:B].to(torch.float32).reshape(-1))
        if draws is not None:
            hard = torch.as_tensor(draws.pop(0), dtype=torch.float32, device=dev).reshape(B, h)
            bvars.append((1.0 - hard).reshape(-1))          # hard = (bvar < probs)
        else:
            bvars.append(torch.rand(B * h, device=dev))
    u = torch.cat(us) if us else torch.zeros(1, device=dev)
    bvar = torch.cat(bvars) if bvars else torch.zeros(1, device=dev)

    # ---- one fused forward + backward on the normalised problem
    Sy = float(V.Sy.reshape(-1)[0])
    Xn = ((x0 - V.mx) * V.iSx).contiguous()
    Yn = ((targets - V.my) / V.Sy).contiguous()
    rw_eff = reg_weight * B / (Sy * Sy)
    key = (id(V), B, tuple(dims), tuple(temps), rw_eff)
    st = _STEPS.get(key)
    if st is None:
        if len(_STEPS) > 8:
            _STEPS.clear()
        # the step computes  mse_n + (rw'/N) reg ; we want  Sy^2 mse_n + reg_weight reg  -> rw'/N = reg_weight / Sy^2
        st = _STEPS[key] = E.BnnStep(dims, temps, rscale, dreg, B, B, reg_weight=rw_eff, device=dev,
                                     loss_kind='mse')
    idx = torch.arange(B, dtype=torch.int32, device=dev)
    gpad, loss = st.loss_grad(Xn, Yn, idx, padded, u, bvar)
    grad = torch.cat([gpad[:-(2 * K + 2)], gpad[-(2 * K + 2):-(K + 2)], gpad[-2:-1]]) * (Sy * Sy)
    cache['step'] += 1
    g = cache['group']
    E.clip_adam(flat, grad, cache['m'], cache['v'], cache['step'], g['lr'], g['betas'], g['eps'], max_norm=None)
    _sync_adam_state(opt, params, cache)
    for dr in drops:
        if isinstance(dr, CDropout):
            dr.p = dr.logit_p.detach().sigmoid()
    if V_target is not None and polyak_averaging > 0:
        tau = polyak_averaging
        with torch.no_grad():
            for p, tp in zip(V.parameters(), V_target.parameters()):
                tp.data.copy_(tau * p.data + (1 - tau) * tp.data)
    V.eval()
    return loss[0] * (Sy * Sy)
