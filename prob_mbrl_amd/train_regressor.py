"""`prob_mbrl.utils.train_regressor` (utils/train_regressor.py:58-165): maximum-likelihood
training of the Bayesian dynamics model.  The iteration body -- forward in train() mode with
concrete dropout, Gaussian log-likelihood, dropout regulariser, backward -- is one device call
(pmbrl_bnn_loss_grad), the optimiser step is the fused Adam of the policy path (pmbrl_clip_adam)."""
import os

import numpy as np
import torch

from . import engine as E
from .algorithms import _adam_flat_state, _sync_adam_state
from .models import CDropout


def iterate_minibatches(N, batchsize):
    """Index stream of utils/train_regressor.py:14-22 (np.random.shuffle every epoch)."""
    while True:
        indices = np.arange(0, max(N, batchsize)) % N
        np.random.shuffle(indices)
        for i in range(0, N, batchsize):
            yield indices[i:i + batchsize]


priority_tree = {}          # utils/train_regressor.py:11: one SumTree per model, kept across calls


def iterate_priority_tree(N, batchsize, tree, warmup_iters=100):
    """utils/train_regressor.py:25-46: uniform minibatches for warmup_iters steps (counting the
    visits), then rows drawn from the priority tree with importance weights (beta 0.4 -> 1).
    Yields (data indices, tree indices, weights or None)."""
    if N > tree.size:
        for i in range(tree.size, N):
            tree.append(i, tree.max_p)
        tree.renormalize()
    it = iterate_minibatches(N, batchsize)
    beta = 0.4
    for _ in range(warmup_iters):
        idxs = next(it)
        tree.counts[idxs] += 1
        tree.max_count = max(tree.max_count, tree.counts[idxs].max())
        yield idxs, idxs + tree.max_size - 1, None
    while True:
        data_idxs, idxs, weights = tree.sample(batchsize, beta=beta)
        beta = min(1.0, beta + 1e-3)
        yield np.array(data_idxs), idxs, weights


def flat_module_parameters(params, owner, key='_pmbrl_flat_all'):
    """Make `params` (in order) views of ONE flat fp32 buffer and return it; redone when the
    views were broken (module.cuda() / .float() / load())."""
    flat = getattr(owner, key, None)
    ok = flat is not None
    if ok:
        off = 0
        for p in params:
            n = p.numel()
            if (p.device != flat.device or p.dtype != torch.float32 or not p.is_contiguous()
                    or p.data_ptr() != flat.data_ptr() + 4 * off):
                ok = False
                break
            off += n
        ok = ok and off == flat.numel()
    if not ok:
        flat = torch.cat([p.detach().reshape(-1).float() for p in params]).contiguous()
        off = 0
        for p in params:
            n = p.numel()
            p.data = flat[off:off + n].view(p.shape)
            off += n
        setattr(owner, key, flat)
    return flat


def _is_diag_gaussian_ll(fn):
    """None, losses.gaussian_log_likelihood, or the bound DiagGaussianDensity.log_prob the example
    scripts pass (examples/deep_pilco_mm.py:224; same formula, models/densities.py:123-144)."""
    if fn is None or getattr(fn, '__name__', '') == 'gaussian_log_likelihood':
        return True
    from .models import DiagGaussianDensity
    return isinstance(getattr(fn, '__self__', None), DiagGaussianDensity) and fn.__name__ == 'log_prob'


def _is_mixture_ll(fn):
    """losses.gaussian_mixture_log_likelihood, or the bound GaussianMixtureDensity.log_prob (same formula,
    losses.py:40-64 / models/densities.py:235-252)."""
    if getattr(fn, '__name__', '') == 'gaussian_mixture_log_likelihood':
        return True
    from .models import GaussianMixtureDensity
    return isinstance(getattr(fn, '__self__', None), GaussianMixtureDensity) and fn.__name__ == 'log_prob'


def train_regressor(model, iters=2000, batchsize=100, resample=True, optimizer=None, log_likelihood=None,
                    reg_weight=1.0, pbar_class=None, summary_writer=None, summary_scope='',
                    decoupled_reg=False, prioritized_sampling=False, priority_eps=1e-3, priority_alpha=0.6,
                    _replay=None, _warmup_iters=100):
    """Same call as the reference.  Offered on the device: the default Gaussian likelihood with a
    plain torch.optim.Adam; decoupled_reg (the regulariser's gradient applied by a separate plain
    SGD step, :133-147) and prioritized_sampling (minibatches from a per-model SumTree with
    importance weights, priorities from the rows' log-likelihoods, :88-131)."""
    from .models import GaussianMixtureDensity
    mixture = isinstance(getattr(model, 'output_density', None), GaussianMixtureDensity)
    if mixture:
        if not _is_mixture_ll(log_likelihood):
            raise NotImplementedError('a GaussianMixtureDensity head is trained with '
                                      'losses.gaussian_mixture_log_likelihood (pass it as log_likelihood)')
    elif not _is_diag_gaussian_ll(log_likelihood):
        raise NotImplementedError('the diagonal-Gaussian and the Gaussian-mixture log-likelihoods are offered on '
                                  'the device path')
    model.train()
    dev = model.mx.device
    if dev.type != 'cuda':
        raise RuntimeError('the model must live on a HIP device (no CPU fallback)')
    Xn = ((model.X - model.mx) * model.iSx).to(torch.float32).contiguous()
    Yn = ((model.Y - model.my) * model.iSy).to(torch.float32).contiguous()
    N = Xn.shape[0]
    print('train_regressor >', 'Dataset size [%d]' % int(N))
    linears, drops, inner = model.model.layer_spec()
    density = inner if inner is not None else model.output_density
    if density is None:
        raise NotImplementedError('a diagonal-Gaussian output density is required')
    dims = [linears[0].in_features] + [l.out_features for l in linears]
    # module parameter order: fc0.weight, fc0.bias, drop0.logit_p, fc1.weight, ...
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
                raise NotImplementedError('BNN training is offered for concrete dropout (CDropout) layers')
    if any(not p.requires_grad for p in params):
        raise NotImplementedError('frozen parameters are not offered on the device path')
    model_params = [p for p in model.parameters() if p.requires_grad]
    if len(model_params) != len(params) or any(a is not b for a, b in zip(model_params, params)):
        raise NotImplementedError('the model has trainable parameters outside Linear / CDropout layers')
    if optimizer is None:
        optimizer = torch.optim.Adam(params, 1e-4)
    flat = flat_module_parameters(params, model)
    cache = _adam_flat_state(optimizer, params, flat)
    if cache is None:
        raise NotImplementedError('BNN training on the device needs a plain torch.optim.Adam over '
                                  'model.parameters()')
    steps = {}
    grad = torch.empty_like(flat)
    sum_h = sum(d for d, t in zip(dims[1:-1], temps) if t > 0)
    u_fixed = None
    tree = None
    if prioritized_sampling:
        from .experience import SumTree
        tree = priority_tree.get(model)
        if tree is None or N > tree.size:
            old = tree
            tree = SumTree(2 * N)
            if old is not None:
                tree.max_p = old.max_p
                tree.counts[:len(old.counts)] = old.counts
            priority_tree[model] = tree
        batches = iterate_priority_tree(N, batchsize, tree, _warmup_iters)
    else:
        batches = ((ix, None, None) for ix in iterate_minibatches(N, batchsize))
    # uniform minibatches: one host-to-device copy of the shuffled indices per epoch, not per step
    epoch_dev = {}

    def device_indices(idx_np):
        if prioritized_sampling:
            return torch.as_tensor(idx_np.astype(np.int32), device=dev)
        base = idx_np.base if idx_np.base is not None else idx_np
        hit = epoch_dev.get('base')
        if hit is not base:
            epoch_dev['base'] = base
            epoch_dev['dev'] = torch.as_tensor(base.astype(np.int32), device=dev)
        off = (idx_np.__array_interface__['data'][0] - base.__array_interface__['data'][0]) // base.itemsize
        return epoch_dev['dev'][off:off + len(idx_np)]
    rng = range(iters + 1)       # the reference runs iters + 1 steps (`if i == iters: break` after the step)
    pbar = pbar_class(rng, total=iters) if pbar_class is not None else rng
    last = None
    # The plain call (what the example scripts issue 2 000 iterations at a time, examples/deep_pilco_mm.py:218-227): whole
    # iterations on the device, two launches each, queued in chunks by ONE library call per chunk (pmbrl_bnn_train_steps) --
    # minibatch indices uploaded per chunk, dropout noise drawn in the kernel from a seed taken from torch's generator
    # (reproducible under torch.manual_seed), Adam's step on the device.  The recorded draws of a test replay go through
    # the same kernels, one iteration per call.
    fused = (not mixture and not prioritized_sampling and not decoupled_reg and sum_h > 0 and
             (resample or _replay is not None) and os.environ.get('PMBRL_BNN_FUSED', '1') != '0')
    if fused:
        g = cache['group']
        step_dev = torch.tensor([int(cache['step'])], dtype=torch.int64, device=dev)
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if _replay is None else 0
        chunk = 1 if _replay is not None else 50
        n_total, done = iters + 1, 0
        pending = []
        it_pbar = iter(pbar)
        while done < n_total:
            # consecutive minibatches of one size (an epoch's last one may be short)
            if not pending:
                pending.append(next(batches)[0])
            M = len(pending[0])
            take = [pending.pop(0)]
            while len(take) < min(chunk, n_total - done):
                nxt = next(batches)[0]
                if len(nxt) != M:
                    pending.append(nxt)
                    break
                take.append(nxt)
            st = steps.get(M)
            if st is None:
                st = steps[M] = E.BnnStep(dims, temps, rscale, dreg, M, N, reg_weight,
                                          max_log_std=float(density.max_log_std), device=dev)
            idx_all = torch.as_tensor(np.stack(take).astype(np.int32), device=dev).contiguous()
            u = bv = None
            if _replay is not None:
                f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev).reshape(-1)  # noqa: E731
                u = torch.cat([f32(a) for a in _replay['u'][done]]).contiguous()
                bv = torch.cat([1.0 - f32(a) for a in _replay['hard'][done]]).contiguous()      # hard = (bvar < probs)
            hist = torch.empty(len(take), 3, dtype=torch.float32, device=dev)
            loss = st.train_steps(Xn, Yn, idx_all, flat, cache['m'], cache['v'], step_dev, g['lr'], g['betas'], g['eps'],
                                  seed=seed, first_step=done, u=u, bvar=bv, loss_hist=hist)
            cache['step'] += len(take)
            last = loss
            if summary_writer is not None:
                names = ['training_loss', 'E_lml', 'reg_loss']
                if summary_scope:
                    names = ['/'.join([summary_scope, n]) for n in names]
                for k, lv in enumerate(hist.tolist()):
                    summary_writer.add_scalar(names[0], lv[0], done + k)
                    summary_writer.add_scalar(names[1], -lv[1], done + k)
                    summary_writer.add_scalar(names[2], lv[2], done + k)
            for k in range(len(take)):
                i = next(it_pbar, None)
                if i is not None and hasattr(pbar, 'set_description') and (i % 50 == 0):
                    pbar.set_description('log-likelihood of data: %f' % (-float(hist[k, 1])))
            done += len(take)
        pbar = ()
    for i in pbar:
        idx_np, tree_idx, w_np = next(batches)
        M = len(idx_np)
        st = steps.get(M)
        if st is None:
            st = steps[M] = E.BnnStep(dims, temps, rscale, dreg, M, N, reg_weight,
                                      max_log_std=float(density.max_log_std), device=dev,
                                      loss_kind='gmm' if mixture else 'nll',
                                      n_components=density.n_components if mixture else 0)
        idx = device_indices(idx_np)
        if _replay is not None:     # tests: the reference's recorded draws of this step
            f32 = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32), device=dev).reshape(-1)  # noqa: E731
            u_fixed = torch.cat([f32(a) for a in _replay['u'][i]])
            bvar = torch.cat([1.0 - f32(a) for a in _replay['hard'][i]])        # hard = (bvar < probs)
        else:
            if resample or u_fixed is None or u_fixed.numel() != M * sum_h:
                u_fixed = torch.rand(M * sum_h, device=dev, dtype=torch.float32)
            bvar = torch.rand(M * sum_h, device=dev, dtype=torch.float32)
        row_w = row_lp = None
        if prioritized_sampling:
            row_lp = torch.empty(M, dtype=torch.float32, device=dev)
            if w_np is not None:
                row_w = torch.as_tensor(np.stack(w_np).astype(np.float32), device=dev).contiguous()
        _, loss = st.loss_grad(Xn, Yn, idx, flat, u_fixed, bvar, grad, row_weight=row_w, row_logprob=row_lp,
                               terms=1 if decoupled_reg else 3)
        cache['step'] += 1
        g = cache['group']
        E.clip_adam(flat, grad, cache['m'], cache['v'], cache['step'], g['lr'], g['betas'], g['eps'],
                    max_norm=None)
        if prioritized_sampling:
            # new priorities from the rows' log-likelihoods (utils/train_regressor.py:117-128)
            a = 2
            p0 = 1 + (a - np.clip(row_lp.cpu().numpy().reshape(-1), -a, a)) / (2 * a)
            pri = (p0 * tree.max_count / tree.counts[tree_idx - tree.max_size + 1] + priority_eps)**priority_alpha
            for ti, pv in zip(tree_idx, pri):
                tree.update(ti, pv)
            tree.renormalize()
        if decoupled_reg:
            # the regulariser's own step: plain SGD with the optimiser's learning rate (:133-147)
            if decoupled_reg and loss is not None:
                loss = loss.clone()
            _, lreg = st.loss_grad(Xn, Yn, idx, flat, u_fixed, bvar, grad, terms=2)
            flat.add_(grad, alpha=-float(g['lr']))
            loss[2] = lreg[2]
        last = loss
        if summary_writer is not None:
            lv = loss.tolist()
            names = ['training_loss', 'E_lml', 'reg_loss']
            if summary_scope:
                names = ['/'.join([summary_scope, n]) for n in names]
            summary_writer.add_scalar(names[0], lv[0], i)
            summary_writer.add_scalar(names[1], -lv[1], i)
            summary_writer.add_scalar(names[2], lv[2], i)
        if hasattr(pbar, 'set_description') and (i % 50 == 0):
            pbar.set_description('log-likelihood of data: %f' % (-float(loss[1])))
    _sync_adam_state(optimizer, params, cache)
    for dr in drops:
        if isinstance(dr, CDropout):
            dr.p = dr.logit_p.detach().sigmoid()
    model.eval()
    return last
