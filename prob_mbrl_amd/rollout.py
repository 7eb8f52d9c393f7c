"""`rollout()` with the reference's call surface (utils/rollout.py:62-163), backed
by the fused HIP kernels.  Autograd sees ONE node per rollout
(`RolloutFunction`), not thousands of aten ops: forward = pmbrl_rollout_fwd,
backward = pmbrl_rollout_bwd (adjoint sweep + dW GEMM)."""
import numpy as np
import torch

from . import engine as E
from . import models as M


# ---------------------------------------------------------------------------
# flat parameter storage: the kernels take ONE pointer per network
# ---------------------------------------------------------------------------
def flat_parameters(linears, owner):
    """Make the Linear weights/biases views of one flat fp32 buffer (torch parameter
    order W0, b0, W1, b1, ...) and return it.  Re-done whenever the views were broken
    (e.g. by module.cuda() / .float() / load())."""
    params = []
    for lin in linears:
        params += [lin.weight, lin.bias]
    flat = getattr(owner, '_pmbrl_flat', None)
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
        owner._pmbrl_flat = flat
    return flat, params


def _scalar_of(module, name):
    """float(module.<name>) without a host-device sync per call: memoised on the buffer's storage
    and version counter."""
    t = getattr(module, name)
    sig = (t.data_ptr(), t._version, str(t.device))
    cache = module.__dict__.setdefault('_pmbrl_scalars', {})
    hit = cache.get(name)
    if hit is None or hit[0] != sig:
        hit = (sig, float(t))
        cache[name] = hit
    return hit[1]


class _MaskCache:
    """bit-packed masks keyed on the identity + version of the float mask tensor"""

    def __init__(self):
        self.store = {}

    def get(self, key, mask, B, gen=0):
        sig = (mask.data_ptr(), mask._version, tuple(mask.shape), B, gen)
        hit = self.store.get(key)
        if hit is None or hit[0] != sig:
            m = mask[:B]
            if m.dtype != torch.float32:
                m = m.float()
            hit = (sig, E.pack_mask(m.contiguous()))
            self.store[key] = hit
        return hit[1]

    def ones(self, key, B, width, device):
        sig = ('ones', B, width, str(device))
        hit = self.store.get(key)
        if hit is None or hit[0] != sig:
            hit = (sig, E.pack_mask(torch.ones(B, width, device=device)))
            self.store[key] = hit
        return hit[1]


_ENGINES = {}
_MASKS = _MaskCache()


def _masked(key, dr, B, w):
    mask = dr.hard_mask(B, w)          # may redraw (and bump the generation) first
    return _MASKS.get(key, mask, B, getattr(dr, '_mask_gen', 0))


def _reward_key(spec):
    return (spec['kind'], spec['expand'], tuple(spec['angle_dims']),
            np.asarray(spec['C'], dtype=np.float64).tobytes(),
            np.asarray(spec['tip_target'], dtype=np.float64).tobytes(), float(spec['norm']),
            float(spec['w']), np.asarray(spec['Q'], dtype=np.float64).tobytes(),
            np.asarray(spec['R'], dtype=np.float64).tobytes())


def get_engine(B, D, U, H, pol_dims, pol_keep, dyn_dims, dyn_keep, spec, mm_states, mm_rewards,
               mm_groups, device, B_global=None, row_offset=0, zmm_per_step=False,
               max_log_std=(E.LOG_MAX_STD, E.LOG_MAX_STD), infer_ns=False, precision=None,
               masks_per_step=(False, False), angle_dims=((), ()), gmm=(0, False), mm_span=None):
    precision = precision or E.get_precision()
    key = (str(device), B, D, U, H, tuple(pol_dims), tuple(pol_keep), tuple(dyn_dims),
           tuple(dyn_keep), _reward_key(spec), bool(mm_states), bool(mm_rewards), mm_groups,
           B_global, row_offset, zmm_per_step, max_log_std, bool(infer_ns), precision, tuple(masks_per_step),
           tuple(angle_dims[0]), tuple(angle_dims[1]), tuple(gmm), tuple(mm_span) if mm_span else None)
    eng = _ENGINES.get(key)
    if eng is None:
        if len(_ENGINES) > 16:
            _ENGINES.clear()
        eng = E.Engine(B, D, U, H, pol_dims, pol_keep, dyn_dims, dyn_keep, spec,
                       mm_states=mm_states, mm_rewards=mm_rewards, mm_groups=mm_groups,
                       device=device, B_global=B_global, row_offset=row_offset,
                       zmm_per_step=zmm_per_step, max_log_std_pol=max_log_std[0],
                       max_log_std_dyn=max_log_std[1], infer_ns=infer_ns, precision=precision,
                       pol_masks_per_step=masks_per_step[0], dyn_masks_per_step=masks_per_step[1],
                       pol_angle_dims=angle_dims[0], dyn_angle_dims=angle_dims[1],
                       dyn_components=gmm[0], gmm_exact_noise_grad=gmm[1], mm_span=mm_span)
        _ENGINES[key] = eng
    return eng


class Bundle:
    """Everything one rollout needs, gathered from the reference-shaped modules."""

    def __init__(self, dynamics, policy, B, H, resample_state_noise, resample_action_noise,
                 mm_states, mm_rewards, mm_groups, z_mm, z_rr, B_global=None, row_offset=0,
                 infer_ns=False, precision=None, resample_policy=False, resample_model=False,
                 mm_span=None, process_group=None):
        if not isinstance(policy, M.Policy) or not isinstance(dynamics, M.DynamicsModel):
            raise TypeError('rollout() needs prob_mbrl_amd.models.Policy / DynamicsModel')
        if dynamics.reward_func is None or not hasattr(dynamics.reward_func, 'spec'):
            raise NotImplementedError('DynamicsModel.reward_func must be a prob_mbrl_amd.rewards '
                                      'module (learned rewards are not offered)')
        plin, pdrop, pdens = policy.model.layer_spec()
        dlin, ddrop, ddens_inner = dynamics.model.layer_spec()
        ddens = dynamics.output_density
        self.gmm = isinstance(ddens, M.GaussianMixtureDensity)
        if not isinstance(pdens, M.DiagGaussianDensity) or ddens_inner is not None or \
                not (isinstance(ddens, M.DiagGaussianDensity) or self.gmm):
            raise NotImplementedError('policy needs a DiagGaussianDensity output_nonlin and the dynamics a '
                                      'DiagGaussianDensity or GaussianMixtureDensity output_density')
        self.pol_flat, self.pol_params = flat_parameters(plin, policy.model)
        self.dyn_flat, _ = flat_parameters(dlin, dynamics.model)
        dev = self.pol_flat.device
        if dev.type != 'cuda':
            raise RuntimeError('prob_mbrl_amd: modules must live on a HIP device (no CPU fallback)')
        if self.dyn_flat.device != dev:
            raise RuntimeError('policy and dynamics must be on the same device')
        self.device = dev
        self.pol_dims = [plin[0].in_features] + [l.out_features for l in plin]
        self.dyn_dims = [dlin[0].in_features] + [l.out_features for l in dlin]
        # angle_dims (models/core.py:233-234,173-174): the networks see [others | sin | cos] of their input
        self.angle_dims = (tuple(int(a) for a in policy.angle_dims.tolist()),
                           tuple(int(a) for a in dynamics.angle_dims.tolist()))
        self.D = self.pol_dims[0] - len(self.angle_dims[0])
        self.U = self.pol_dims[-1] // 2
        n_dyn_in = self.D + self.U + len(self.angle_dims[1])
        self.n_comp = ddens.n_components if self.gmm else 0
        n_dyn_out = (2 * self.D + 1) * self.n_comp + 1 if self.gmm else 2 * self.D
        if self.dyn_dims[0] != n_dyn_in or self.dyn_dims[-1] != n_dyn_out:
            raise NotImplementedError('dynamics must map [x|u] (%d inputs with its angle_dims) to 2*|x| outputs; '
                                      'a learned reward head is not offered' % n_dyn_in)
        if any(a >= self.D for a in self.angle_dims[1]):
            raise NotImplementedError('angle_dims of the dynamics model must be state dimensions')
        self.B, self.H = B, H
        if getattr(dynamics.reward_func, 'bind_action_dim', None):
            dynamics.reward_func.bind_action_dim(self.U)
        self.spec = dynamics.reward_func.spec(self.D)
        self.pol_keep = [dr.keep_prob() if dr is not None else 1.0 for dr in pdrop]
        self.dyn_keep = [dr.keep_prob() if dr is not None else 1.0 for dr in ddrop]
        # per-unit Bernoulli rates: 1 / p_j folded into row j of the Linear in front of the dropout (models.BDropout.
        # unit_inv_keep); the kernels then see scaled COPIES of the parameters and the gradient is scaled back
        self.pol_unit = self._unit_rows(pdrop, self.pol_dims, dev)
        self.dyn_unit = self._unit_rows(ddrop, self.dyn_dims, dev)
        self.unit_scaled = bool(self.pol_unit)
        # masks (frozen unless resampled by the caller through .resample())
        self.pol_bits, self.dyn_bits = [], []

        def step_masks(dr, w):
            """A fresh mask per time step (resample=True in the module's forward: models/modules.py:55-58 for
            Bernoulli dropout -- drawn and not stored --, :134-139,155-157 for concrete dropout in eval mode --
            new uniform noise, a hard sample of its concrete probabilities, stored), bit rows [H * B, nt] drawn by
            one device launch (pmbrl_draw_masks)."""
            return dr.step_mask_bits(H, B, w)

        for i, dr in enumerate(pdrop):
            w = self.pol_dims[i + 1]
            key = (id(policy), 'p', i)
            if dr is not None and resample_policy:
                self.pol_bits.append(step_masks(dr, w))
            elif dr is None and resample_policy:
                self.pol_bits.append(E.pack_mask(torch.ones(H * B, w, device=dev)))
            else:
                self.pol_bits.append(_MASKS.ones(key, B, w, dev) if dr is None else _masked(key, dr, B, w))
        for i, dr in enumerate(ddrop):
            w = self.dyn_dims[i + 1]
            key = (id(dynamics), 'd', i)
            if dr is not None and resample_model:
                self.dyn_bits.append(step_masks(dr, w))
            elif dr is None and resample_model:
                self.dyn_bits.append(E.pack_mask(torch.ones(H * B, w, device=dev)))
            else:
                self.dyn_bits.append(_MASKS.ones(key, B, w, dev) if dr is None else _masked(key, dr, B, w))
        # output noise: frozen buffer, or a fresh draw per step (models/densities.py:111-119)
        if resample_action_noise:
            self.z_pol = torch.randn(H, B, self.U, device=dev)
            pdens.z.data = self.z_pol[-1]
        else:
            self.z_pol = pdens.frozen_noise(B, False)
        self.z_pi = self.u_cat = None
        if self.gmm:
            # mixture head (models/densities.py:213-231): frozen Gumbel noise; a component draw and new Gaussian
            # noise at every step, whatever resample_state_noise says (the reference's own behaviour)
            self.z_pi = ddens.frozen_gumbel(B, resample_state_noise)
            forced = getattr(ddens, '_forced_draws', None)      # tests: replay recorded draws
            if forced is not None:
                self.z_dyn, self.u_cat = (f.to(device=dev, dtype=torch.float32).contiguous() for f in forced)
            else:
                self.z_dyn = torch.randn(H, B, self.D, device=dev)
                self.u_cat = torch.rand(H, B, device=dev)
            ddens.z_normal.data = self.z_dyn[-1]
        elif resample_state_noise:
            self.z_dyn = torch.randn(H, B, self.D, device=dev)
            ddens.z.data = self.z_dyn[-1]
        else:
            self.z_dyn = ddens.frozen_noise(B, False)
        self.max_log_std = (_scalar_of(pdens, 'max_log_std'), _scalar_of(ddens, 'max_log_std'))
        f = lambda t, n: t.detach().reshape(-1).float().expand(n).contiguous()  # noqa: E731
        self.mx, self.iSx = f(dynamics.mx, n_dyn_in), f(dynamics.iSx, n_dyn_in)
        self.my, self.Sy = f(dynamics.my, self.D), f(dynamics.Sy, self.D)
        self.scale, self.bias = f(policy.scale, self.U), f(policy.bias, self.U)
        # moment matching noise
        self.mm_states, self.mm_rewards, self.mm_groups = mm_states, mm_rewards, mm_groups
        Bg = B_global if B_global is not None else B
        self.zmm_per_step = False
        self.z_mm = self.z_rr = None
        if mm_states:
            if z_mm is None:   # utils/rollout.py:58-59: fresh noise every step
                self.zmm_per_step = True
                z_mm = torch.randn(H, Bg, self.D, device=dev)
            self.z_mm = z_mm
        if mm_rewards:
            if z_rr is None:
                self.zmm_per_step = True
                z_rr = torch.randn(H, Bg, 1, device=dev)
            self.z_rr = z_rr
        if self.zmm_per_step and ((mm_states and self.z_mm.dim() != 3) or
                                  (mm_rewards and self.z_rr.dim() != 3)):
            raise NotImplementedError('pass both z_mm and z_rr, or neither')
        self.engine = get_engine(B, self.D, self.U, H, self.pol_dims, self.pol_keep, self.dyn_dims,
                                 self.dyn_keep, self.spec, mm_states, mm_rewards, mm_groups, dev,
                                 B_global=B_global, row_offset=row_offset,
                                 zmm_per_step=self.zmm_per_step, max_log_std=self.max_log_std,
                                 infer_ns=infer_ns and (mm_states or mm_rewards), precision=precision,
                                 masks_per_step=(bool(resample_policy), bool(resample_model)),
                                 angle_dims=self.angle_dims,
                                 gmm=(self.n_comp, bool(getattr(ddens, 'exact_noise_grad', False))),
                                 mm_span=mm_span if (mm_states or mm_rewards) else None)
        if mm_span and (mm_states or mm_rewards):
            # groups spread over the ranks of process_group: their per-step statistics cross devices
            if process_group is None:
                raise ValueError('mm_span needs the process_group whose ranks share the groups')
            self.engine.attach_collective(process_group)

    @staticmethod
    def _unit_rows(drops, dims, dev):
        """[(offset of W_l, offset of b_l, out, in, 1 / p [out])] of the layers followed by a per-unit BDropout, in the
        flat parameter order W0, b0, W1, b1, ..."""
        rows, off = [], 0
        for l in range(len(dims) - 1):
            n_in, n_out = dims[l], dims[l + 1]
            dr = drops[l] if l < len(drops) else None
            inv = dr.unit_inv_keep() if dr is not None and hasattr(dr, 'unit_inv_keep') else None
            if inv is not None:
                if inv.numel() != n_out:
                    raise ValueError('BDropout rate has %d entries, its layer %d units' % (inv.numel(), n_out))
                rows.append((off, off + n_out * n_in, n_out, n_in, inv.to(dev)))
            off += n_out * n_in + n_out
        return rows

    @staticmethod
    def scale_unit_rows(flat, rows):
        """flat with row j of every listed W (and b_j) multiplied by 1 / p_j: the parameters the kernels see, and --
        applied to the gradient they return -- dL/d(the module's parameters)."""
        if not rows:
            return flat
        out = flat.clone()
        for ow, ob, n_out, n_in, inv in rows:
            out[ow:ow + n_out * n_in].view(n_out, n_in).mul_(inv[:, None])
            out[ob:ob + n_out].mul_(inv)
        return out

    def forward(self, x0, out=None):
        if self.pol_unit or self.dyn_unit:
            return self.engine.forward(x0, self.scale_unit_rows(self.pol_flat, self.pol_unit),
                                       self.scale_unit_rows(self.dyn_flat, self.dyn_unit), self.mx, self.iSx, self.my,
                                       self.Sy, self.scale, self.bias, self.pol_bits, self.dyn_bits,
                                       self.z_pol, self.z_dyn, self.z_mm, self.z_rr, out=out, z_pi=self.z_pi,
                                       u_cat=self.u_cat)
        return self.engine.forward(x0, self.pol_flat, self.dyn_flat, self.mx, self.iSx, self.my,
                                   self.Sy, self.scale, self.bias, self.pol_bits, self.dyn_bits,
                                   self.z_pol, self.z_dyn, self.z_mm, self.z_rr, out=out, z_pi=self.z_pi,
                                   u_cat=self.u_cat)


class RolloutFunction(torch.autograd.Function):
    """One autograd node for the whole H-step rollout."""

    @staticmethod
    def forward(ctx, bundle, x0, *pol_params):
        eng = bundle.engine
        dev = bundle.device
        out = (torch.empty((eng.H + 1, eng.B, eng.D), device=dev),
               torch.empty((eng.H, eng.B, eng.U), device=dev),
               torch.empty((eng.H, eng.B, 1), device=dev))
        S, A, R = bundle.forward(x0, out=out)
        ctx.bundle = bundle
        ctx.generation = eng.generation
        ctx.set_materialize_grads(False)
        return S, A, R

    @staticmethod
    def backward(ctx, gS, gA, gR):
        bundle = ctx.bundle
        eng = bundle.engine
        if eng.generation != ctx.generation:
            raise RuntimeError('stale rollout graph: another rollout ran on this plan before '
                               'backward() (the stashes were overwritten)')
        if gR is None:
            gR = torch.zeros((eng.H, eng.B), device=bundle.device)
        # (after a late failure the caller holds the first n steps only; the engine's status word makes
        #  the adjoint sweep and the dW GEMM stop there too, so the non-finite stashes of the steps
        #  >= n are never touched -- utils/rollout.py:154-157)
        want_x0 = ctx.needs_input_grad[1]
        agn_out = getattr(bundle, 'agn_out', None)
        g, gx0, agn = eng.backward(gR, grad_states=gS, grad_actions=gA, want_x0=want_x0,
                                   want_agn=agn_out is not None)
        if agn_out is not None:
            agn_out.append(agn)
        g = bundle.scale_unit_rows(g, bundle.pol_unit) if bundle.pol_unit else g.clone()
        grads, off = [], 0
        for p in bundle.pol_params:
            n = p.numel()
            grads.append(g[off:off + n].view(p.shape))
            off += n
        return (None, gx0) + tuple(grads)


def get_z_rnd(z, i, shape, device=None):
    """utils/rollout.py:53-59 (kept for API parity; the kernels index the buffer themselves)."""
    if z is not None:
        idxs = torch.arange(i, i + shape[0], device=z.device) % shape[0]
        return z[idxs]
    return torch.randn(*shape, device=device)


def rollout(states, dynamics, policy, steps, resample_model=False, resample_policy=False,
            resample_state_noise=True, resample_action_noise=True, mm_states=False,
            mm_rewards=False, infer_noise_variables=False, z_mm=None, z_rr=None, mm_groups=None,
            breaking_condition=None, on_step=None, on_pol_eval=None, **kwargs):
    """Trajectory distribution (s_0, a_0, r_0, s_1, ...) of `policy` on `dynamics` from the
    given particles: returns [states (steps+1 x [B,D]), actions (steps x [B,U]),
    rewards (steps x [B,1])], connected to policy.parameters() (and to `states`) for autograd.
    Same arguments as the reference (utils/rollout.py:62-79)."""
    if callable(on_pol_eval):
        raise NotImplementedError('on_pol_eval needs a per-step Python hook; not offered')
    B = states.shape[0]
    B_global, row_offset = kwargs.pop('B_global', None), kwargs.pop('row_offset', 0)
    agn_out = kwargs.pop('action_grad_norms_out', None)
    precision = kwargs.pop('precision', None)
    # moment-matching groups spread over the ranks of process_group (not in the reference, which is single-process):
    # mm_span = (rows of a group over all ranks, this rank's first row inside each group, ranks, this rank)
    mm_span, process_group = kwargs.pop('mm_span', None), kwargs.pop('process_group', None)
    # sharded callers (mc_pilco with world > 1): the group whose ranks must keep the SAME horizon and take the same
    # retry / raise decision -- a rank that truncates, retries or raises alone leaves the others inside the next
    # collective (the gradient all-reduce, the row gathers of CVaR and prioritised replay)
    agree_group = kwargs.pop('agree_group', None)

    def _agree(n_local, failed_local):
        if agree_group is None:
            return n_local, failed_local
        import torch.distributed as dist
        if dist.get_world_size(agree_group) == 1:
            return n_local, failed_local
        v = torch.tensor([n_local, -int(failed_local)], dtype=torch.int32, device=bundle.device)
        if dist.get_backend(agree_group) == 'nccl':
            dist.all_reduce(v, op=dist.ReduceOp.MIN, group=agree_group)
        else:
            host = v.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.MIN, group=agree_group)
            v = host
        n_all, failed_all = int(v[0]), int(v[1]) < 0
        if n_all < n_local:
            # what the adjoint sweep, the dW GEMM and the loss of THIS rank read: stop where the slowest rank stopped
            bundle.engine.status[0:1].copy_(torch.tensor([n_all], dtype=torch.int32))
        return n_all, failed_all

    while True:
        bundle = Bundle(dynamics, policy, B, int(steps), resample_state_noise, resample_action_noise,
                        mm_states, mm_rewards, mm_groups, z_mm, z_rr, B_global=B_global,
                        row_offset=row_offset, infer_ns=bool(infer_noise_variables), precision=precision,
                        resample_policy=bool(resample_policy), resample_model=bool(resample_model),
                        mm_span=mm_span, process_group=process_group)
        # a list that backward() appends the [H, B] matrix of ||dL/da_t|| to (prioritised replay)
        bundle.agn_out = agn_out
        x0 = states.to(device=bundle.device, dtype=torch.float32)
        S, A, R = RolloutFunction.apply(bundle, x0, *bundle.pol_params)
        n = bundle.engine.valid_steps()
        failed = n < steps
        n, failed = _agree(n, failed)
        if agree_group is None and mm_span and E.safe_precision(bundle.engine.info['precision']) is not None:
            # groups spread over ranks: a rank retrying alone would leave the others inside a collective -- the ranks
            # agree (every one of them makes this call), and retry together if any of them failed
            failed = bundle.engine.any_rank(failed)
        retry = E.safe_precision(bundle.engine.info['precision']) if failed else None
        if retry is None or resample_state_noise or resample_action_noise or (mm_states and z_mm is None) \
                or resample_model or resample_policy or bundle.gmm:
            break          # (fresh noise was drawn: a retry would be a different rollout)
        # a failure under fp16 pieces may be their range, not the rollout: decide on the bf16 path
        precision = retry
    if n < steps:
        # utils/rollout.py:154-157: keep a truncated horizon if enough steps succeeded
        if n <= 5:
            raise RuntimeError('rollout failed at step %d (non-finite state or non-positive '
                               'Cholesky pivot in moment matching)' % n)
    states_l = list(S[:n + 1].unbind(0))
    actions_l = list(A[:n].unbind(0))
    rewards_l = list(R[:n].unbind(0))
    if callable(breaking_condition) or callable(on_step):
        # the hooks see the trajectory prefix step by step, as in the reference loop
        for t in range(n):
            traj = list(zip(states_l[:t + 1], actions_l[:t + 1], rewards_l[:t + 1]))
            if callable(breaking_condition) and breaking_condition(traj):
                states_l, actions_l, rewards_l = states_l[:t + 2], actions_l[:t + 1], rewards_l[:t + 1]
                break
            if callable(on_step):
                on_step(traj)
    return [states_l, actions_l, rewards_l]
