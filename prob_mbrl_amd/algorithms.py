"""`mc_pilco()` with the reference's call surface (algorithms/mc_pilco.py:13-267).

Two execution paths, same semantics:
 * fused (default): rollout forward -> loss -> adjoint sweep -> dW GEMM ->
   [gradient all-reduce] -> clip + Adam are C-ABI calls on flat buffers; no
   autograd graph, no torch.nn op, one host sync per iteration (status + loss).
   Used when the optimiser is a plain torch.optim.Adam over the policy's Linear
   parameters and no option needs autograd (CVaR, regulariser, value_func,
   prioritized_replay).
 * autograd: `rollout()`'s single autograd node + torch's loss / clip / optimiser
   for everything else the reference signature allows.  value_func is evaluated by
   pmbrl_mlp_forward and differentiated by pmbrl_mlp_grad_input; the priorities of
   prioritized_replay come from pmbrl_rollout_bwd's action_grad_norms output.
"""
import math
import traceback
from collections import defaultdict

import numpy as np
import torch

from . import engine as E
from . import rollout as RO
from ._lib import PmbrlError
from .utils import tile

policy_update_counter = defaultdict(lambda: 0)   # algorithms/mc_pilco.py:8
x0_tree = None          # SumTree(2**20) of start states, built on first use (mc_pilco.py:9)
episode_counter = 0     # episodes of `exp` already in x0_tree (mc_pilco.py:10)


def _discount_fn(discount, steps):
    """algorithms/mc_pilco.py:46-50."""
    if discount is None:
        return lambda i: 1.0 / steps
    if not callable(discount):
        factor = discount
        return lambda i: factor**i
    return discount


def _adam_flat_state(opt, params, flat):
    """If `opt` is a plain Adam over exactly `params`, return (m, v, step, group) with the
    per-parameter state re-pointed at flat buffers (views), else None."""
    if type(opt) is not torch.optim.Adam or len(opt.param_groups) != 1:
        return None
    g = opt.param_groups[0]
    if g.get('weight_decay', 0) != 0 or g.get('amsgrad', False) or g.get('maximize', False):
        return None
    # frozen tensors (e.g. the reward module's constants inside DynamicsModel.parameters()) sit in
    # the group but never receive a gradient: Adam skips them, so do we
    gp = [p for p in g['params'] if p.requires_grad]
    if len(gp) != len(params) or any(a is not b for a, b in zip(gp, params)):
        return None
    cache = getattr(opt, '_pmbrl_flat', None)
    if cache is not None and cache['flat_ptr'] == flat.data_ptr():
        st = opt.state.get(params[0], {})
        if 'step' in st:   # torch may have stepped this optimiser in between
            cache['step'] = int(st['step'])
        return cache
    m = torch.zeros_like(flat)
    v = torch.zeros_like(flat)
    step, off = 0, 0
    for p in params:
        n = p.numel()
        st = opt.state.get(p, {})
        if 'exp_avg' in st:
            m[off:off + n] = st['exp_avg'].reshape(-1)
            v[off:off + n] = st['exp_avg_sq'].reshape(-1)
            step = int(st['step']) if not torch.is_tensor(st['step']) else int(st['step'].item())
        off += n
    cache = dict(m=m, v=v, step=step, group=g, flat_ptr=flat.data_ptr())
    opt._pmbrl_flat = cache
    return cache


def _sync_adam_state(opt, params, cache):
    """Expose the flat Adam state through opt.state (views) so state_dict()/later
    opt.step() calls keep working."""
    off = 0
    for p in params:
        n = p.numel()
        st = opt.state[p]
        st['step'] = torch.tensor(float(cache['step']))
        st['exp_avg'] = cache['m'][off:off + n].view(p.shape)
        st['exp_avg_sq'] = cache['v'][off:off + n].view(p.shape)
        off += n


def mc_pilco(init_states, dynamics, policy, steps, opt=None, exp=None, opt_iters=1000,
             value_func=None, pegasus=True, mm_states=False, mm_rewards=False, mm_groups=None,
             maximize=True, clip_grad=1.0, cvar_eps=0.0, reg_weight=0.0, discount=None,
             on_rollout=None, on_iteration=None, step_idx_to_sample=None, init_state_noise=0.0,
             resampling_period=99, prioritized_replay=False, priority_alpha=0.6,
             priority_eps=1e-8, init_priority_beta=1.0, priority_beta_increase=0.0, debug=False,
             rollout_kwargs={}, process_group=None, frozen_noise=None, progress=False):
    """Monte-Carlo PILCO policy search.  Mutates `policy` parameters and `opt` state in place.

    Extra keyword arguments (not in the reference): `process_group` -- a torch.distributed
    group over which the particle rows are sharded (this rank passes ITS rows; the flat policy
    gradient is all-reduced over RCCL before the identical clip + Adam on every rank);
    `frozen_noise` -- dict(z_mm=..., z_rr=...) to inject captured PEGASUS noise (tests);
    `progress` -- print the reference's tqdm-style line every 50 iterations."""
    global policy_update_counter, x0_tree, episode_counter
    if prioritized_replay and x0_tree is None:
        from .experience import SumTree
        x0_tree = SumTree(2**20)
    dynamics.eval()
    policy.train()
    H = int(steps)
    disc = _discount_fn(discount, H)
    if opt is None:
        opt = torch.optim.Adam(filter(lambda p: p.requires_grad, policy.parameters()))
    dev = next(policy.parameters()).device
    if dev.type != 'cuda':
        raise RuntimeError('prob_mbrl_amd.mc_pilco needs the modules on a HIP device')
    D = init_states.shape[-1]
    N_particles = init_states.shape[0]
    world, rank = 1, 0
    mm_span = None      # one moment-matching group over the particles of all ranks (sharded runs, mm_groups=None)
    if process_group is not None:
        import torch.distributed as dist
        world, rank = dist.get_world_size(process_group), dist.get_rank(process_group)
        if world > 1 and (mm_states or mm_rewards) and mm_groups is None:
            # one Gaussian over ALL particles (the examples' default): the per-step sufficient statistics are
            # summed over the ranks (pmbrl_config.mm_span_rows) -- fitting one Gaussian per rank
            # instead would silently be different mathematics.  Needs the same particle count on every rank.
            cnt = torch.tensor([N_particles, -N_particles], dtype=torch.int64, device=dev)
            dist.all_reduce(cnt, op=dist.ReduceOp.MAX, group=process_group)
            if int(cnt[0]) != -int(cnt[1]):
                raise ValueError('moment matching over one global group on a sharded run needs the same number of '
                                 'particles on every rank')
            mm_span = (N_particles * world, rank * N_particles, world, rank)
    Bg = N_particles * world
    z_mm = torch.randn(H + Bg, D, device=dev)
    z_rr = torch.randn(H + Bg, 1, device=dev)
    if frozen_noise is not None:
        z_mm = frozen_noise['z_mm'].to(dev, torch.float32)
        z_rr = frozen_noise['z_rr'].to(dev, torch.float32)

    def resample():
        if frozen_noise is not None:
            return
        seed = torch.randint(2**32, [1])
        if world > 1:   # every rank must draw different masks: offset the seed by the rank
            seed = seed + 7919 * rank
        dynamics.resample(seed=seed)
        policy.resample(seed=seed)
        if value_func is not None:
            value_func.resample(seed=seed)
        z_mm.normal_()
        z_rr.normal_()
        if mm_span is not None:
            # one group over all ranks: one cyclic noise buffer, as one process would hold it
            dist.broadcast(z_mm, dist.get_global_rank(process_group, 0), group=process_group)
            dist.broadcast(z_rr, dist.get_global_rank(process_group, 0), group=process_group)

    resample()
    x0 = init_states
    n_opt_steps = policy_update_counter[policy]
    cvar = cvar_eps > -1.0 and cvar_eps < 1.0 and cvar_eps != 0
    # sharded runs: CVaR is a row mask on dL/dr once the returns of ALL ranks are known, the regulariser touches only
    # the (replicated) parameters -- both fit the fused path; one process keeps the reference's own
    # sequence of autograd operations for them
    need_autograd = (((cvar or reg_weight > 0) and world == 1)
                     or value_func is not None or prioritized_replay or bool(rollout_kwargs))
    replay = None
    if prioritized_replay:      # algorithms/mc_pilco.py:80-84
        replay = dict(idxs=None, weights=None, beta=init_priority_beta, eps=priority_eps,
                      alpha=priority_alpha, tree=x0_tree, N=N_particles, world=world, rank=rank,
                      group=process_group)
    gamma_full = [float(disc(i)) for i in range(H)]
    sign = -1.0 if maximize else 1.0

    # Pipelined fused path (single process, no on_rollout hook): the optimiser step is decided on
    # the device from the rollout's status word (pmbrl_clip_adam_guarded) and the host reads that
    # word one iteration late, from a pinned snapshot queued right behind the forward sweep.  The
    # host then never waits for a whole iteration: while it prepares iteration i+1 the device still
    # has the adjoint sweep, the dW GEMM and the Adam step of iteration i to run.
    pipelined = (world == 1) and not callable(on_rollout) and not need_autograd
    # arithmetic of the sweeps: the library default, until a rollout fails under fp16 pieces -- whose
    # range, not the rollout, may be what failed: the rest of this call then runs on the bf16 pieces
    prec = dict(name=None)
    pending = None          # (i, event, pinned status, loss, S, A, R) of the iteration in flight
    pipe = dict(snap=[torch.empty(1, dtype=torch.int32).pin_memory() for _ in range(2)],
                ev=[torch.cuda.Event() for _ in range(2)], step_dev=None,
                loss=[torch.zeros(1, dtype=torch.float32, device=dev) for _ in range(2)]) if pipelined else None

    # a rollout that failed after more than 5 steps is optimised on its truncated horizon
    # (utils/rollout.py:154-157); earlier failures skip the step (algorithms/mc_pilco.py:122-131)
    min_steps = min(H, 6)

    pend_prec = [None]      # arithmetic of the iteration in flight

    def settle(pend):
        """Host side of a pipelined iteration, once its status word has arrived."""
        nonlocal n_opt_steps
        j, ev, snap, loss_j, S_j, A_j, R_j = pend
        ev.synchronize()
        n_valid = min(int(snap[0]), H)
        if n_valid < H and prec['name'] is None and E.safe_precision(pend_prec[0]) is not None:
            # (the device skipped or truncated this step on its own; no resampling: same random numbers,
            #  wider exponent range from now on)
            prec['name'] = E.safe_precision(pend_prec[0])
            print('rollout failed at step %d under %s arithmetic (iteration %d): continuing with %s' %
                  (n_valid, pend_prec[0], j, prec['name']))
            if n_valid < min_steps:
                return
        elif n_valid < min_steps:
            # algorithms/mc_pilco.py:122-131: report, draw new random numbers; the device already
            # skipped the optimiser step
            print('RuntimeError: rollout failed at step %d (iteration %d)' % (n_valid, j))
            resample()
            return
        n_opt_steps += 1
        if (progress and j % 50 == 0) or callable(on_iteration):
            # fresh tensors, like the reference hands out: the engine's buffers are overwritten by the
            # next iteration
            states = list(S_j[:n_valid + 1].clone().unbind(0))
            actions, rewards = list(A_j[:n_valid].clone().unbind(0)), list(R_j[:n_valid].clone().unbind(0))
            if progress and j % 50 == 0:
                msg = 'Pred. Cumm. rewards: %f' if maximize else 'Pred. Cumm. costs: %f'
                print((msg % float(torch.stack(rewards).sum(0).mean())) + ' [{0}]'.format(len(rewards)))
            if callable(on_iteration):
                on_iteration(j, loss_j.clone(), states, actions, rewards, disc)     # (the buffer is reused two iterations on)

    for i in range(opt_iters):
        if pending is not None:
            settle(pending)
            pending = None
        if not pegasus or n_opt_steps % resampling_period == 0:
            resample()
        x0_ = x0
        if mm_groups is not None and x0_.shape[0] == mm_groups:
            x0_ = tile(x0_, int(N_particles / mm_groups))
        x0_ = x0_.to(dev, torch.float32)
        if not (isinstance(init_state_noise, float) and init_state_noise == 0.0):
            x0_ = x0_ + torch.as_tensor(init_state_noise, device=dev) * torch.randn_like(x0_)
        rk = dict(rollout_kwargs)
        queued = False
        try:
            dist_kw = dict(B_global=Bg if world > 1 else None, row_offset=rank * N_particles if world > 1 else 0,
                           mm_span=mm_span, process_group=process_group if mm_span else None)
            bundle = RO.Bundle(dynamics, policy, x0_.shape[0], H, not pegasus, not pegasus,
                               mm_states, mm_rewards, mm_groups, z_mm if pegasus else None,
                               z_rr if pegasus else None, precision=prec['name'], **dist_kw)
            # (per-unit dropout rates: the kernels run on scaled copies of the parameters -- the device-side Adam step
            #  would update the wrong numbers; the autograd form scales the gradient back)
            cache = None if (need_autograd or bundle.unit_scaled) else _adam_flat_state(opt, bundle.pol_params,
                                                                                         bundle.pol_flat)
            if cache is None:
                loss, states, actions, rewards = _autograd_iteration(
                    x0_, dynamics, policy, H, opt, pegasus, mm_states, mm_rewards, mm_groups, z_mm,
                    z_rr, maximize, clip_grad, cvar_eps, reg_weight, disc, on_rollout, i, rk,
                    process_group, world, value_func, replay, dist_kw)
            elif pipelined:
                eng = bundle.engine
                if pipe['step_dev'] is None:
                    pipe['step_dev'] = torch.tensor([cache['step']], dtype=torch.int64, device=dev)
                    pipe['cache'] = cache
                S, A, R = bundle.forward(x0_)
                pend_prec[0] = eng.info['precision']
                k = i & 1
                pipe['snap'][k].copy_(eng.status[0:1], non_blocking=True)
                pipe['ev'][k].record()
                gw = _loss_weights(eng, gamma_full, sign, Bg, dev)
                # (the loss: formed by the backward call's gradient reduction, written by its optimiser launch --
                #  pmbrl_adam::loss_out_d; two buffers: the host reads iteration i's one iteration late)
                loss = pipe['loss'][k][0]
                grp = cache['group']
                # adjoint sweep, dW GEMM, then reduction + global norm + clip + Adam, taken on the device only if the
                # rollout completed (pmbrl_rollout_bwd_adam)
                eng.backward(gw, adam=dict(params=bundle.pol_flat, exp_avg=cache['m'], exp_avg_sq=cache['v'],
                                           step=pipe['step_dev'], lr=grp['lr'], betas=grp['betas'], eps=grp['eps'],
                                           max_norm=clip_grad, expect=min_steps, loss_out=pipe['loss'][k]))
                pending = (i, pipe['ev'][k], pipe['snap'][k], loss, S, A, R)
                queued = True
            else:
                eng = bundle.engine
                def agreed_steps(e):
                    """Valid steps of the sweep just run -- on a sharded run the MIN over the ranks: every rank must take
                    the same branch and the same horizon, a failure anywhere truncates (or fails) the rollout
                    everywhere, like one process would."""
                    n = e.valid_steps()           # the one host sync of the iteration
                    if world > 1:
                        import torch.distributed as dist
                        nv = torch.tensor([n], dtype=torch.int32, device=dev)
                        dist.all_reduce(nv, op=dist.ReduceOp.MIN, group=process_group)
                        if int(nv[0]) < n:
                            e.status[0:1].copy_(nv)           # what the adjoint sweep and the loss read
                            n = int(nv[0])
                    return n
                S, A, R = bundle.forward(x0_)
                n_valid = agreed_steps(eng)
                if n_valid < H and prec['name'] is None and E.safe_precision(eng.info['precision']) is not None:
                    # fp16 pieces: their range may be what failed -- decide on the bf16 / fp32 path.  (n_valid is the
                    # ranks' minimum, so all of them switch together and stay switched.)
                    prec['name'] = E.safe_precision(eng.info['precision'])
                    bundle = RO.Bundle(dynamics, policy, x0_.shape[0], H, not pegasus, not pegasus, mm_states,
                                       mm_rewards, mm_groups, z_mm if pegasus else None, z_rr if pegasus else None,
                                       precision=prec['name'], **dist_kw)
                    eng = bundle.engine
                    S, A, R = bundle.forward(x0_)
                    n_valid = agreed_steps(eng)
                if n_valid < min_steps:
                    raise RuntimeError('rollout failed at step %d' % n_valid)
                gw = _loss_weights(eng, gamma_full, sign, Bg, dev)
                if cvar:
                    # algorithms/mc_pilco.py:146-154 over the rows of all ranks: the returns are gathered in rank (=
                    # row) order, the quantile is the one process' quantile, every rank keeps its own rows' mask
                    gcol = torch.tensor(gamma_full[:n_valid], dtype=torch.float32, device=dev)
                    ret = sign * (R[:n_valid, :, 0] * gcol[:, None]).sum(0)
                    rd = _gather_rows(ret, process_group, world).cpu().numpy()
                    if cvar_eps > 0:
                        q = np.quantile(rd, cvar_eps)
                        sel, n_sel = ret < float(q), int((rd < q).sum())
                    else:
                        q = np.quantile(rd, -cvar_eps)
                        sel, n_sel = ret > float(q), int((rd > q).sum())
                    gfull = torch.tensor(gamma_full, dtype=torch.float64) * (sign / max(n_sel, 1))
                    gw = (gfull.float().to(dev)[:, None] * sel.float()[None, :]).contiguous()
                loss = eng.weighted_sum(R, gw)[0]
                if world > 1:
                    loss = loss.reshape(1).clone()
                    dist.all_reduce(loss, group=process_group)
                    loss = loss[0]
                if callable(on_rollout) or callable(on_iteration):
                    # the hooks get tensors of their own (the engine's buffers are reused by the next
                    # iteration; the reference hands out fresh tensors)
                    Sc, Ac, Rc = S.clone(), A.clone(), R.clone()
                else:
                    Sc, Ac, Rc = S, A, R
                states, actions, rewards = list(Sc[:n_valid + 1].unbind(0)), list(Ac[:n_valid].unbind(0)), \
                    list(Rc[:n_valid].unbind(0))
                if callable(on_rollout):
                    on_rollout(i, states, actions, rewards, disc)
                g, _, _ = eng.backward(gw)
                if eng.info.get('mm_grid') or eng.info.get('mm_parts', 1) > 1:
                    # sweeps whose workgroups meet at barriers: a barrier that timed out (workgroups not co-resident
                    # after all -- another process on the device) leaves a garbage gradient and says so in status[1]
                    bad = eng.sweep_failed()
                    if world > 1:
                        import torch.distributed as dist
                        fl = torch.tensor([1 if bad else 0], dtype=torch.int32, device=dev)
                        dist.all_reduce(fl, op=dist.ReduceOp.MAX, group=process_group)
                        bad = bool(int(fl[0]))
                    if bad:
                        raise RuntimeError('adjoint sweep: a barrier between workgroups timed out')
                if world > 1:
                    from .distributed import grad_allreduce, p2p_check
                    grad_allreduce(process_group, dev)(g)     # RCCL on the compute stream (C ABI)
                    # (peer-to-peer transport: a wait that timed out anywhere -- this gradient, or the per-step
                    #  statistics of groups spread over the ranks -- fails the iteration on every rank)
                    p2p_check(process_group, dev)
                if reg_weight > 0:
                    # algorithms/mc_pilco.py:193-194: a function of the parameters alone, the same on every rank
                    policy.zero_grad()
                    regl = policy.regularization_loss()
                    regl.backward()
                    g.add_(torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1)
                                      for p in bundle.pol_params]), alpha=float(reg_weight))
                    loss = loss + reg_weight * regl.detach()
                cache['step'] += 1
                grp = cache['group']
                E.clip_adam(bundle.pol_flat, g, cache['m'], cache['v'], cache['step'], grp['lr'],
                            grp['betas'], grp['eps'], max_norm=clip_grad)
        except NotImplementedError:      # an option the device path does not offer: not a failed rollout
            raise
        except RuntimeError:
            # algorithms/mc_pilco.py:122-131: print, draw new random numbers, skip the step
            traceback.print_exc()
            print('RuntimeError')
            resample()
            opt.zero_grad()
            continue
        if not queued:
            n_opt_steps += 1
            if progress and i % 50 == 0:
                msg = 'Pred. Cumm. rewards: %f' if maximize else 'Pred. Cumm. costs: %f'
                print((msg % float(torch.stack(rewards).sum(0).mean())) + ' [{0}]'.format(len(rewards)))
            if callable(on_iteration):
                on_iteration(i, loss, states, actions, rewards, disc)
        # new initial states (algorithms/mc_pilco.py:222-263)
        if exp is not None:
            n_draw = mm_groups if mm_groups is not None else N_particles
            if prioritized_replay:
                # grow the tree with the episodes it has not seen (mc_pilco.py:224-231)
                if exp.n_samples() > x0_tree.size:
                    for idx in range(episode_counter, exp.n_episodes()):
                        for x in torch.as_tensor(np.asarray(exp.states[idx])):
                            x0_tree.append(x, x0_tree.max_p)
                            x0_tree.renormalize()
                    episode_counter = exp.n_episodes()
                # sharded run: every rank holds the same tree and uses the same GLOBAL sample (drawn by rank 0), then keeps
                # its slice of the start states and importance weights; the priorities of the whole sample are updated
                # on every rank from the gathered norms (_update_priorities), so the replicas of the tree stay identical
                if world > 1:
                    # ONE draw for all ranks: rank 0's (the replicas' numpy generators are not assumed in step -- a
                    # single user draw on one rank would silently give the ranks different samples, weights and,
                    # through _update_priorities, different trees)
                    import torch.distributed as dist
                    box = [None]
                    if rank == 0:
                        xs, idxs0, w = x0_tree.sample(n_draw * world, beta=replay['beta'])
                        box = [(np.asarray(idxs0), np.asarray(w))]
                    dist.broadcast_object_list(box, src=dist.get_global_rank(process_group, 0), group=process_group)
                    replay['idxs'], w = box[0]
                    if rank != 0:
                        xs = x0_tree.replay_sample(replay['idxs'])
                else:
                    xs, replay['idxs'], w = x0_tree.sample(n_draw * world, beta=replay['beta'])
                # (sic) max, not min: beta never drops below 1 (mc_pilco.py:240-241)
                replay['beta'] = max(1.0, replay['beta'] + priority_beta_increase)
                lo_r, hi_r = rank * n_draw, (rank + 1) * n_draw
                x0 = torch.stack([torch.as_tensor(x) for x in xs[lo_r:hi_r]]).to(dev, torch.float32)
                # (the reference multiplies the [B, 1] returns with the [B] weights -- a [B, B] product, mc_pilco.py:156-158:
                #  every row meets the weights of ALL sampled states, so a rank keeps the whole weight vector)
                replay['weights'] = torch.as_tensor(np.stack(w)).to(dev, torch.float32)
            else:
                x0 = exp.sample_states(n_draw, timestep=step_idx_to_sample).to(dev, torch.float32)
                init_states = x0
        else:
            x0 = init_states.detach()

    if pending is not None:
        settle(pending)
    if pipe is not None and pipe['step_dev'] is not None:
        pipe['cache']['step'] = int(pipe['step_dev'].item())
    cache = getattr(opt, '_pmbrl_flat', None)
    if cache is not None and type(opt) is torch.optim.Adam:
        _sync_adam_state(opt, [p for p in opt.param_groups[0]['params'] if p.requires_grad], cache)
    policy.eval()
    dynamics.eval()
    policy_update_counter[policy] = n_opt_steps


def _gather_rows(t, group, world):
    """The rows of every rank's `t` ([n, ...], the same n on every rank) in rank order, on every rank -- a tensor
    collective (no pickling, no host round trip beyond what the backend needs)."""
    import torch.distributed as dist
    parts = [torch.empty_like(t) for _ in range(world)]
    if dist.get_backend(group) == 'nccl':
        dist.all_gather(parts, t.contiguous(), group=group)
        return torch.cat(parts, 0)
    host = [torch.empty(t.shape, dtype=t.dtype) for _ in range(world)]     # (gloo groups: the CPU-transport tests)
    dist.all_gather(host, t.detach().cpu().contiguous(), group=group)
    return torch.cat(host, 0).to(t.device)


_GW_CACHE = {}


def _loss_weights(eng, gamma, sign, B_global, dev):
    """dL/dr[t,b] = -/+ gamma_t / B  (algorithms/mc_pilco.py:134-144,190), cached per shape."""
    key = (id(eng), tuple(gamma), sign, B_global)
    gw = _GW_CACHE.get(key)
    if gw is None:
        if len(_GW_CACHE) > 8:
            _GW_CACHE.clear()
        col = torch.tensor(gamma, dtype=torch.float64) * (sign / B_global)
        gw = col.float().to(dev)[:, None].expand(eng.H, eng.B).contiguous()
        _GW_CACHE[key] = gw
    return gw


def _update_priorities(replay, m_norms, mm_groups):
    """New priorities of the sampled start states from the per-step ||dL/da_t|| of their
    particles (mc_pilco.py:163-182; the reference collects the same norms with tensor hooks on
    actions[t], here they are an output of the adjoint sweep)."""
    tree, idxs = replay['tree'], replay['idxs']
    world = replay.get('world', 1)
    if world > 1:      # [H, B_local] -> [H, B_global] in rank (= row) order
        m_norms = _gather_rows(m_norms.t().contiguous(), replay['group'], world).t()
    if mm_groups is not None:
        m_norms = m_norms.reshape(-1, mm_groups * world, int(replay['N'] / mm_groups)).mean(-1)
    scores = m_norms.mean(0).detach().cpu().numpy() / tree.counts[idxs - tree.max_size + 1]
    priorities = (scores + replay['eps'])**replay['alpha']
    for idx, p in zip(idxs, priorities):
        tree.update(idx, p)
    tree.renormalize()


def _autograd_iteration(x0_, dynamics, policy, H, opt, pegasus, mm_states, mm_rewards, mm_groups,
                        z_mm, z_rr, maximize, clip_grad, cvar_eps, reg_weight, disc, on_rollout, i,
                        rollout_kwargs, process_group, world, value_func=None, replay=None, dist_kw=None):
    """The reference loop body on top of the autograd rollout.  On a sharded run (world > 1) this rank's rows are a
    slice of one global batch: the loss is the mean over the GLOBAL batch (CVaR: over the rows its global quantile
    keeps), the flat gradient is summed over the ranks before the clip, the regulariser -- a function of the
    replicated parameters -- counts once."""
    dkw = {}
    if world > 1:
        import torch.distributed as dist
        dkw = {k: v for k, v in (dist_kw or {}).items() if v is not None}
        # every rank keeps the horizon of the slowest one and retries / raises with it (rollout(): agree_group)
        dkw['agree_group'] = process_group
    policy.zero_grad()
    opt.zero_grad()
    weighted = replay is not None and replay['idxs'] is not None
    norms_out = [] if weighted else None
    states, actions, rewards = RO.rollout(
        x0_, dynamics, policy, H, resample_state_noise=not pegasus,
        resample_action_noise=not pegasus, mm_states=mm_states, mm_rewards=mm_rewards,
        z_mm=z_mm if pegasus else None, z_rr=z_rr if pegasus else None, mm_groups=mm_groups,
        action_grad_norms_out=norms_out, **dkw, **rollout_kwargs)
    if callable(on_rollout):
        on_rollout(i, states, actions, rewards, disc)
    discounted = torch.stack([r * disc(t) for t, r in enumerate(rewards)])
    if value_func is not None:
        # terminal-value bootstrap (mc_pilco.py:136-140); dV/ds_H comes from pmbrl_mlp_grad_input
        # and enters the adjoint sweep as grad_states[H]
        Vend = value_func(states[-1], resample=False, return_samples=True)
        discounted = torch.cat([discounted, disc(H) * Vend.unsqueeze(0)], 0)
    returns = -discounted.sum(0) if maximize else discounted.sum(0)
    n_global = returns.shape[0] * world
    if cvar_eps > -1.0 and cvar_eps < 1.0 and cvar_eps != 0:
        rd = returns.detach()
        rd_all = (_gather_rows(rd, process_group, world) if world > 1 else rd).cpu().numpy()
        if cvar_eps > 0:
            q = np.quantile(rd_all, cvar_eps)
            returns, n_global = returns[rd < q], int((rd_all < q).sum())
        else:
            q = np.quantile(rd_all, -cvar_eps)
            returns, n_global = returns[rd > q], int((rd_all > q).sum())
    if weighted:
        # importance-sampling weights, broadcast exactly as the reference does ([B,1] * [n],
        # mc_pilco.py:156-158)
        returns = returns * replay['weights']
    if world == 1:
        loss = returns.mean()
        if reg_weight > 0:
            loss = loss + reg_weight * policy.regularization_loss()
        loss.backward()
    else:
        if weighted and returns.dim() == 2:
            n_global = n_global * returns.shape[1]     # (the reference's [B, 1] * [B] broadcast: a mean over [B, B])
        loss = returns.sum() / max(n_global, 1)
        if reg_weight > 0:
            loss = loss + (reg_weight / world) * policy.regularization_loss()
        loss.backward()
        from .distributed import grad_allreduce
        params = [p for p in policy.parameters() if p.requires_grad]
        flat = torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(-1) for p in params])
        tot = torch.cat([flat, loss.detach().reshape(1)])
        grad_allreduce(process_group, tot.device)(tot)
        from .distributed import p2p_check
        p2p_check(process_group, tot.device)
        off = 0
        for p in params:
            p.grad = tot[off:off + p.numel()].reshape(p.shape).clone()
            off += p.numel()
        loss = tot[-1]
    if weighted:
        _update_priorities(replay, norms_out[0][:len(actions)], mm_groups)
    if clip_grad is not None:
        torch.nn.utils.clip_grad_norm_(policy.parameters(), clip_grad)
    opt.step()
    return loss.detach(), states, actions, rewards
