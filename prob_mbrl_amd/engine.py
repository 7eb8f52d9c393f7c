"""Thin Python wrapper over one pmbrl plan: owns the workspace and trajectory
tensors (torch tensors are device storage only) and issues the C-ABI calls on
torch's current HIP stream."""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib

LOG_MAX_STD = math.log(5.0)   # reference models/densities.py:75

# Arithmetic of the hidden-width GEMMs of the sweep kernels (include/pmbrl.h, PMBRL_PREC_*):
# 'f32' exact fp32 MFMA, 'split' bf16 pieces on the bf16 matrix cores (fp32-equivalent forward sweep,
# two-piece adjoint).  Module default, overridable per call and by the environment.
_PRECISIONS = {'f32': _lib.PREC_F32, 'split': _lib.PREC_SPLIT, 'split_f16': _lib.PREC_SPLIT_F16}
_default_precision = [None]


def set_precision(name):
    """Default arithmetic of plans created from now on: 'f32' or 'split'."""
    assert name in _PRECISIONS, name
    _default_precision[0] = name


def get_precision():
    import os
    return _default_precision[0] or os.environ.get('PMBRL_PRECISION', 'split_f16')


def safe_precision(name):
    """The arithmetic to retry a failed rollout with before the failure is believed: the fp16 pieces of
    'split_f16' overflow beyond +-65504 (a hidden activation that large is reported as a non-finite
    step), the bf16 pieces of 'split' have fp32's range.  None: `name` has no such hazard."""
    return 'split' if name == 'split_f16' else None


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32c(t, device):
    """float32, contiguous, on device (no copy when already so)."""
    if t.dtype != torch.float32 or t.device != device or not t.is_contiguous():
        t = t.to(device=device, dtype=torch.float32).contiguous()
    return t


def pack_mask(mask):
    """{0,1} float mask [B,h] (any row stride) -> int16 tensor [B, ceil(h/16)] of
    little-endian bit rows (pmbrl_pack_mask)."""
    lib = _lib.load()
    assert mask.is_cuda and mask.dim() == 2 and mask.dtype == torch.float32
    if mask.stride(1) != 1:
        mask = mask.contiguous()
    B, h = mask.shape
    bits = torch.empty((B, (h + 15) // 16), dtype=torch.int16, device=mask.device)
    _lib.check(lib.pmbrl_pack_mask(_stream(), _ptr(mask), B, h, mask.stride(0),
                                   _ptr(bits)), 'pmbrl_pack_mask')
    return bits


def draw_masks(kind, seed, offset, param, temp, rows, h, u=None, v=None, aux=None, want=('u', 'hard')):
    """Dropout masks drawn on the device as bit rows [rows, ceil(h/16)] (pmbrl_draw_masks).  kind: 'bernoulli'
    (param = keep probabilities) or 'concrete' (param = logit_p, eval-mode hard sample).  aux = (row0, n): also
    return the float uniforms / hard samples / probabilities of those rows (dict with the names in `want`)."""
    lib = _lib.load()
    param = param.detach().reshape(-1).float().contiguous()
    assert param.is_cuda and param.numel() in (1, h)
    dev = param.device
    bits = torch.empty((rows, (h + 15) // 16), dtype=torch.int16, device=dev)
    a0, an = aux if aux is not None else (0, 0)
    outs = {k: (torch.empty((an, h), dtype=torch.float32, device=dev) if (an and k in want) else None)
            for k in ('u', 'hard', 'probs')}
    for t in (u, v):
        assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.shape == (rows, h))
    _lib.check(lib.pmbrl_draw_masks(_stream(), 0 if kind == 'bernoulli' else 1, int(seed) & (2**64 - 1),
                                    int(offset) & (2**64 - 1), _ptr(param), param.numel(), float(temp), rows, h,
                                    _ptr(u), _ptr(v), _ptr(bits), a0, an, _ptr(outs['u']), _ptr(outs['hard']),
                                    _ptr(outs['probs'])), 'pmbrl_draw_masks')
    return bits, outs


class _MlpCall:
    """Marshalled arguments of one stand-alone network evaluation (kept alive for the backward)."""

    def __init__(self, x, params_flat, dims, keep, mask_bits, z, in_shift, in_iscale, out_scale,
                 out_shift, sq_scale, sq_bias, max_log_std):
        self.lib = _lib.load()
        assert x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.is_contiguous()
        B = x.shape[0]
        nl = len(dims) - 1
        assert x.shape[1] == dims[0] and dims[-1] % 2 == 0
        self.x, self.B, self.n_out, self.dims = x, B, dims[-1] // 2, list(dims)
        call = _lib.MlpCall()
        call.B = B
        call.net.n_layers = nl
        for i, d in enumerate(dims):
            call.net.dims[i] = int(d)
        for l in range(nl - 1):
            call.net.keep[l] = float(keep[l])
        call.max_log_std = float(max_log_std)
        self.call = call
        nbytes = self.lib.pmbrl_mlp_workspace_bytes(C.byref(call))
        if nbytes == 0:
            msg = self.lib.pmbrl_last_error()
            raise ValueError('pmbrl_mlp_workspace_bytes: %s' % (msg.decode() if msg else 'bad shape'))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=x.device)
        self.masks = (C.c_void_p * max(nl - 1, 1))()
        self.keepalive = []
        for l in range(nl - 1):
            m = mask_bits[l] if mask_bits is not None else None
            if m is not None:
                assert m.is_cuda and m.dtype == torch.int16 and m.shape == (B, (dims[l + 1] + 15) // 16)
                m = m.contiguous()
                self.keepalive.append(m)
            self.masks[l] = m.data_ptr() if m is not None else None

        def vec(t, n):
            if t is None:
                return None
            t = t.detach().to(device=x.device, dtype=torch.float32).reshape(-1).contiguous()
            if t.numel() == 1 and n > 1:
                t = t.expand(n).contiguous()
            assert t.numel() == n
            return t

        self.in_shift, self.in_iscale = vec(in_shift, dims[0]), vec(in_iscale, dims[0])
        self.out_scale, self.out_shift = vec(out_scale, self.n_out), vec(out_shift, self.n_out)
        self.sq_scale, self.sq_bias = vec(sq_scale, self.n_out), vec(sq_bias, self.n_out)
        if z is not None:
            assert z.is_cuda and z.dtype == torch.float32 and z.shape == (B, self.n_out)
            z = z.contiguous()
        self.z = z
        self.pf = params_flat.detach().contiguous()

    @staticmethod
    def _p(t):
        return C.c_void_p(t.data_ptr()) if t is not None else None

    def _common(self):
        p = self._p
        return [p(self.x), p(self.pf), self.masks, p(self.z), p(self.in_shift), p(self.in_iscale),
                p(self.out_scale), p(self.out_shift), p(self.sq_scale), p(self.sq_bias)]

    def forward(self, want):
        out = {k: torch.empty((self.B, self.n_out), dtype=torch.float32, device=self.x.device) for k in want}
        p = self._p
        _lib.check(self.lib.pmbrl_mlp_forward(_stream(), C.byref(self.call), p(self.ws), *self._common(),
                                              p(out.get('sample')), p(out.get('mean')), p(out.get('log_std'))),
                   'pmbrl_mlp_forward')
        return out

    def grad_input(self, g_sample=None, g_mean=None, g_log_std=None):
        gx = torch.empty_like(self.x)
        p = self._p
        gs = [None if g is None else g.to(torch.float32).contiguous() for g in (g_sample, g_mean, g_log_std)]
        _lib.check(self.lib.pmbrl_mlp_grad_input(_stream(), C.byref(self.call), p(self.ws), *self._common(),
                                                 p(gs[0]), p(gs[1]), p(gs[2]), p(gx)), 'pmbrl_mlp_grad_input')
        return gx


class _MlpFunction(torch.autograd.Function):
    """Differentiable with respect to the input rows only (the network is a constant here)."""

    @staticmethod
    def forward(ctx, x, call, want):
        ctx.call, ctx.want = call, want
        out = call.forward(want)
        return tuple(out[k] for k in want)

    @staticmethod
    def backward(ctx, *grads):
        g = dict(zip(ctx.want, grads))
        gx = ctx.call.grad_input(g.get('sample'), g.get('mean'), g.get('log_std'))
        return gx, None, None


def mlp_forward(x, params_flat, dims, keep, mask_bits=None, z=None, in_shift=None, in_iscale=None,
                out_scale=None, out_shift=None, sq_scale=None, sq_bias=None, max_log_std=math.log(5.0),
                want=('sample',)):
    """Stand-alone evaluation of one Bayesian MLP with a diagonal-Gaussian head on the rows of x
    (pmbrl_mlp_forward; models/core.py:169-187, 221-248).  dims = [n_in, h..., 2*n_out];
    mask_bits: per hidden layer an int16 bit-row tensor (pack_mask) or None.
    Returns a dict with the requested outputs among 'sample', 'mean', 'log_std' ([B, n_out]).
    If x requires grad the outputs are differentiable with respect to x (pmbrl_mlp_grad_input)."""
    want = tuple(want)
    xc = x.contiguous()
    call = _MlpCall(xc.detach(), params_flat, dims, keep, mask_bits, z, in_shift, in_iscale, out_scale,
                    out_shift, sq_scale, sq_bias, max_log_std)
    if x.requires_grad and torch.is_grad_enabled():
        outs = _MlpFunction.apply(xc, call, want)
        return dict(zip(want, outs))
    return call.forward(want)


class BnnStep:
    """Loss + gradient of one BNN training minibatch on the device (pmbrl_bnn_loss_grad;
    utils/train_regressor.py:113-131).  Flat parameter / gradient order = the module's:
    W0, b0, [logit_p0], W1, b1, [logit_p1], ..., W_L, b_L."""

    def __init__(self, dims, temperature, reg_scale, drop_reg, M, N, reg_weight=1.0,
                 max_log_std=LOG_MAX_STD, device=None, loss_kind='nll', n_components=0):
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else 'cuda:0')
        cfg = _lib.BnnConfig()
        cfg.M, cfg.N = int(M), int(N)
        nl = len(dims) - 1
        cfg.net.n_layers = nl
        for i, d in enumerate(dims):
            cfg.net.dims[i] = int(d)
        for l in range(nl - 1):
            cfg.temperature[l] = float(temperature[l]) if temperature[l] else 0.0
            cfg.reg_scale[l] = float(reg_scale[l])
            cfg.drop_reg[l] = float(drop_reg[l])
        cfg.max_log_std = float(max_log_std)
        cfg.reg_weight = float(reg_weight)
        # 'gmm': mixture-of-Gaussians NLL (losses.py:40-64), head (2 n_out + 1) n_components + 1 wide
        cfg.loss_kind = {'nll': 0, 'mse': 1, 'gmm': 2}[loss_kind]
        cfg.n_components = int(n_components) if loss_kind == 'gmm' else 0
        self.M, self.N, self.dims = int(M), int(N), list(dims)
        self.drop_widths = [dims[l + 1] for l in range(nl - 1) if cfg.temperature[l] > 0]
        self.sum_h = sum(self.drop_widths)
        self.plan = C.c_void_p()
        _lib.check(self.lib.pmbrl_bnn_plan_create(C.byref(cfg), self.device.index or 0, C.byref(self.plan)),
                   'pmbrl_bnn_plan_create')
        self.n_params = int(self.lib.pmbrl_bnn_plan_n_params(self.plan))
        self.ws = torch.empty(self.lib.pmbrl_bnn_plan_workspace_bytes(self.plan), dtype=torch.uint8,
                              device=self.device)
        self.loss = torch.zeros(3, dtype=torch.float32, device=self.device)

    def __del__(self):
        try:
            if self.plan:
                self.lib.pmbrl_bnn_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def loss_grad(self, Xn, Yn, idx, params, u, bvar, grad=None, row_weight=None, row_logprob=None, terms=3):
        """Xn [N, n_in], Yn [N, n_out] normalised dataset; idx int32 [M]; params flat fp32;
        u, bvar: fp32 [M * sum_h] (per dropout layer a block [M, h_l]).  Optional: row_weight [M]
        (importance weights of the rows' log-likelihoods), row_logprob [M] (out: the rows'
        log-likelihoods), terms (1 likelihood, 2 regulariser, 3 both).  Returns (grad, loss[3])."""
        for t in (Xn, Yn, params):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        assert idx.is_cuda and idx.dtype == torch.int32 and idx.numel() == self.M
        assert params.numel() == self.n_params
        if self.sum_h and (terms & 1):
            assert u.numel() == self.M * self.sum_h and bvar.numel() == self.M * self.sum_h
        for t in (row_weight, row_logprob):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.numel() == self.M and t.is_contiguous())
        if grad is None:
            grad = torch.empty_like(params)
        _lib.check(self.lib.pmbrl_bnn_loss_grad_ex(self.plan, _stream(), _ptr(self.ws), _ptr(Xn), _ptr(Yn),
                                                   _ptr(idx), _ptr(params), _ptr(u), _ptr(bvar), _ptr(grad),
                                                   _ptr(self.loss), _ptr(row_weight), _ptr(row_logprob),
                                                   int(terms)), 'pmbrl_bnn_loss_grad_ex')
        return grad, self.loss

    def train_steps(self, Xn, Yn, idx_all, params, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
                    seed=0, first_step=0, u=None, bvar=None, loss_hist=None):
        """Whole training iterations (loss, gradient, torch.optim.Adam's step: utils/train_regressor.py:113-131) in two
        launches each, all queued by this one call (pmbrl_bnn_train_steps).  idx_all int32 [n_steps, M]; params /
        exp_avg / exp_avg_sq flat fp32, updated in place; step: int64 device tensor [1] (Adam's step count, advanced on
        the device).  u, bvar: recorded dropout draws [n_steps, M * sum_h], or None: drawn in the kernel from `seed`
        (counter first_step + i).  loss_hist: optional fp32 [n_steps, 3].  Returns loss[3] of the last iteration."""
        for t in (Xn, Yn, params, exp_avg, exp_avg_sq):
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
        assert idx_all.is_cuda and idx_all.dtype == torch.int32 and idx_all.is_contiguous() and idx_all.numel() % self.M == 0
        n_steps = idx_all.numel() // self.M
        assert n_steps >= 1 and params.numel() == self.n_params == exp_avg.numel() == exp_avg_sq.numel()
        assert step.is_cuda and step.dtype == torch.int64 and step.numel() == 1
        for t in (u, bvar):
            assert t is None or (t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and
                                 t.numel() == n_steps * self.M * self.sum_h)
        assert loss_hist is None or (loss_hist.is_cuda and loss_hist.dtype == torch.float32 and loss_hist.numel() == 3 * n_steps)
        _lib.check(self.lib.pmbrl_bnn_train_steps(self.plan, _stream(), _ptr(self.ws), _ptr(Xn), _ptr(Yn), _ptr(idx_all),
                                                  n_steps, _ptr(params), _ptr(exp_avg), _ptr(exp_avg_sq), _ptr(step),
                                                  float(lr), float(betas[0]), float(betas[1]), float(eps),
                                                  int(seed) & (2 ** 64 - 1), int(first_step), _ptr(u), _ptr(bvar),
                                                  _ptr(self.loss), _ptr(loss_hist)), 'pmbrl_bnn_train_steps')
        return self.loss


def make_reward_struct(spec, D, U):
    """spec: dict(kind, expand, angle_dims, C [k,De], tip_target [k], norm, w,
    Q [k,k], R [U,U]) with numpy / float entries."""
    r = _lib.Reward()
    r.kind = _lib.REWARD_EXP if spec['kind'] == 'exp' else _lib.REWARD_NEG
    r.expand = 1 if spec['expand'] else 0
    ad = [int(a) for a in spec['angle_dims']] if spec['expand'] else []
    r.n_angle = len(ad)
    for i, a in enumerate(ad):
        r.angle_dims[i] = a
    Cm = np.asarray(spec['C'], dtype=np.float32)
    k, De = Cm.shape
    assert De == D + len(ad), 'reward C has %d columns, expected %d' % (De, D + len(ad))
    assert k <= _lib.MAX_TIP and De <= _lib.MAX_DIM
    r.k = k
    for i, v in enumerate(Cm.reshape(-1)):
        r.C[i] = float(v)
    for i, v in enumerate(np.asarray(spec['tip_target'], dtype=np.float32).reshape(-1)):
        r.tip_target[i] = float(v)
    r.norm = float(spec['norm'])
    r.w = float(spec['w'])
    Q = np.asarray(spec['Q'], dtype=np.float32).reshape(k, k)
    for i, v in enumerate(Q.reshape(-1)):
        r.Q[i] = float(v)
    R = np.asarray(spec['R'], dtype=np.float32).reshape(U, U)
    for i, v in enumerate(R.reshape(-1)):
        r.R[i] = float(v)
    return r


class Engine:
    """One (device, shape) plan.  forward() must precede backward()."""

    def __init__(self, B, D, U, H, pol_dims, pol_keep, dyn_dims, dyn_keep,
                 reward_spec, mm_states=False, mm_rewards=False, mm_groups=None,
                 device=None, B_global=None, row_offset=0, rows_per_wg_hint=0,
                 max_log_std_pol=LOG_MAX_STD, max_log_std_dyn=LOG_MAX_STD, zmm_per_step=False,
                 force_generic=False, no_shaped=False, infer_ns=False, precision=None,
                 pol_masks_per_step=False, dyn_masks_per_step=False, pol_angle_dims=(), dyn_angle_dims=(),
                 dyn_components=0, gmm_exact_noise_grad=False, mm_span=None):
        self.lib = _lib.load()
        if not torch.cuda.is_available():
            raise RuntimeError('prob_mbrl_amd needs a HIP device (no CPU fallback)')
        self.device = torch.device(device if device is not None else
                                   'cuda:%d' % torch.cuda.current_device())
        cfg = _lib.Config()
        cfg.B, cfg.D, cfg.U, cfg.H = B, D, U, H
        cfg.B_global = B_global if B_global is not None else B
        cfg.row_offset = row_offset
        cfg.flags = ((_lib.FLAG_MM_STATES if mm_states else 0) |
                     (_lib.FLAG_MM_REWARDS if mm_rewards else 0) |
                     (_lib.FLAG_ZMM_PER_STEP if zmm_per_step else 0) |
                     (_lib.FLAG_INFER_NS if infer_ns else 0) |
                     (_lib.FLAG_POL_MASKS_PER_STEP if pol_masks_per_step else 0) |
                     (_lib.FLAG_DYN_MASKS_PER_STEP if dyn_masks_per_step else 0) |
                     (_lib.FLAG_GMM_EXACT_NOISE_GRAD if gmm_exact_noise_grad else 0) |
                     (_lib.FLAG_FORCE_GENERIC if force_generic else 0) |
                     (_lib.FLAG_NO_SHAPED if no_shaped else 0))
        cfg.mm_groups = int(mm_groups) if mm_groups else 0
        cfg.max_log_std_pol = max_log_std_pol
        cfg.max_log_std_dyn = max_log_std_dyn
        for mlp, dims, keep in ((cfg.pol, pol_dims, pol_keep), (cfg.dyn, dyn_dims, dyn_keep)):
            mlp.n_layers = len(dims) - 1
            assert mlp.n_layers <= _lib.MAX_LAYERS
            for i, d in enumerate(dims):
                mlp.dims[i] = int(d)
            for i in range(_lib.MAX_LAYERS):
                mlp.keep[i] = float(keep[i]) if i < len(keep) else 1.0
        cfg.reward = make_reward_struct(reward_spec, D, U)
        cfg.rows_per_wg_hint = rows_per_wg_hint
        cfg.precision = _PRECISIONS[precision if precision is not None else get_precision()]
        # angle_dims of the policy / the dynamics model (models/core.py:233-234,173-174)
        pol_angle_dims, dyn_angle_dims = [int(a) for a in pol_angle_dims], [int(a) for a in dyn_angle_dims]
        if len(pol_angle_dims) > _lib.MAX_ANGLE or len(dyn_angle_dims) > _lib.MAX_ANGLE:
            raise ValueError('at most %d angle dims per network' % _lib.MAX_ANGLE)
        cfg.n_pol_angle, cfg.n_dyn_angle = len(pol_angle_dims), len(dyn_angle_dims)
        for i, a in enumerate(pol_angle_dims):
            cfg.pol_angle_dims[i] = a
        for i, a in enumerate(dyn_angle_dims):
            cfg.dyn_angle_dims[i] = a
        self.n_dyn_in = D + U + len(dyn_angle_dims)
        # GaussianMixtureDensity dynamics head (models/densities.py:151-259): components (0 / 1: diagonal Gaussian)
        self.n_comp = int(dyn_components) if dyn_components and int(dyn_components) > 1 else 0
        cfg.dyn_components = self.n_comp
        # moment-matching groups spread over ranks: (rows of a group over all ranks, this rank's first row in
        # each group, ranks, this rank) -- pmbrl_config.mm_span_*; needs attach_collective() before forward()
        self.mm_span = tuple(int(v) for v in mm_span) if mm_span else None
        if self.mm_span:
            cfg.mm_span_rows, cfg.mm_span_offset, cfg.mm_span_ranks, cfg.mm_span_rank = self.mm_span
        self._coll = None
        self.cfg = cfg
        self.B, self.D, self.U, self.H = B, D, U, H
        self.n_pol_layers = len(pol_dims) - 1
        self.n_dyn_layers = len(dyn_dims) - 1
        self.mask_rows = (H * B if pol_masks_per_step else B, H * B if dyn_masks_per_step else B)
        plan = C.c_void_p()
        _lib.check(self.lib.pmbrl_plan_create(C.byref(cfg), self.device.index or 0,
                                              C.byref(plan)), 'pmbrl_plan_create')
        self.plan = plan
        info = (C.c_int32 * _lib.INFO_COUNT)()
        _lib.check(self.lib.pmbrl_plan_info(plan, info), 'pmbrl_plan_info')
        self.info = dict(rows_per_wg=info[0], n_wg=info[1], row_tiles=info[2],
                         lds_bytes=info[3], n_pol_params=info[4], n_dyn_params=info[5],
                         dw_splits=info[6], mm_mode=info[7], LD=info[8], dw_blocks=info[9],
                         fast=info[10], stages=(info[11] >> 4, info[11] & 15), mm_grid=info[12],
                         precision={_lib.PREC_SPLIT: 'split', _lib.PREC_SPLIT_F16: 'split_f16'}.get(info[13], 'f32'),
                         dw_pipe=info[14], mm_parts=info[15], reg=info[16], replay=info[17], inplace=info[18])
        self.n_pol_params = info[4]
        self.n_dyn_params = info[5]
        ws_bytes = self.lib.pmbrl_plan_workspace_bytes(plan)
        self.ws_bytes = ws_bytes
        self.workspace = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=self.device)
        off = (-self.workspace.data_ptr()) % 256
        self._ws_ptr = C.c_void_p(self.workspace.data_ptr() + off)
        dev = self.device
        self.states = torch.empty((H + 1, B, D), dtype=torch.float32, device=dev)
        self.actions = torch.empty((H, B, U), dtype=torch.float32, device=dev)
        self.rewards = torch.empty((H, B, 1), dtype=torch.float32, device=dev)
        # [0]: valid steps of the last forward sweep, [1]: failure flag of the last adjoint sweep
        self.status = torch.zeros(2, dtype=torch.int32, device=dev)
        self.grad_flat = torch.empty(self.n_pol_params, dtype=torch.float32, device=dev)
        self._traj = (self.states, self.actions, self.rewards)
        self._inputs = None
        self._keep = None
        self.generation = 0    # bumped by every forward(); backward must match

    def __del__(self):
        try:
            if getattr(self, 'plan', None):
                self.lib.pmbrl_plan_destroy(self.plan)
                self.plan = None
        except Exception:
            pass

    def attach_collective(self, group):
        """The statistics exchange of moment-matching groups spread over the ranks of `group` (pmbrl_config.
        mm_span_rows): RCCL through the C ABI on the compute stream when the group runs on it (pmbrl_plan_set_comm),
        otherwise a host-staged torch.distributed all-reduce handed to the library as a callback
        (pmbrl_plan_set_collective: gloo groups -- the CPU-transport tests with both ranks on one device)."""
        if self._coll is not None and self._coll[0] is group:
            return
        ws = self.workspace
        from .distributed import P2PComm
        if isinstance(group, P2PComm):
            # one-shot peer-to-peer transport (pmbrl_p2p.hip): attached inside the library, one kernel per exchange on
            # the compute stream, nothing of the host language in the per-step loop
            _lib.check(self.lib.pmbrl_plan_set_p2p(self.plan, group.p2p), 'pmbrl_plan_set_p2p')
            self._coll = (group, group)
            return
        if callable(group):
            # a transport of the caller's: group(view) sums the fp64 device tensor `view` over the ranks in place
            # (tests: ranks as threads of one process)
            on_device, custom = False, group
        else:
            import torch.distributed as dist
            from .distributed import get_comm, get_p2p, p2p_wanted
            if p2p_wanted():         # PMBRL_P2P=1: the one-shot peer-to-peer transport for the per-step statistics
                p2p = get_p2p(group, self.device)
                _lib.check(self.lib.pmbrl_plan_set_p2p(self.plan, p2p.p2p), 'pmbrl_plan_set_p2p')
                self._coll = (group, p2p)
                return
            comm = get_comm(group, self.device)
            if comm is not None:
                _lib.check(self.lib.pmbrl_plan_set_comm(self.plan, comm.comm), 'pmbrl_plan_set_comm')
                self._coll = (group, comm)
                return
            on_device, custom = dist.get_backend(group) == 'nccl', None

        def allreduce(ctx, stream, buf, n):
            try:
                off = int(buf) - ws.data_ptr()
                view = ws[off:off + 8 * n].view(torch.float64)
                if custom is not None:
                    custom(view)
                elif on_device:
                    dist.all_reduce(view, group=group)
                else:
                    host = view.cpu()          # (orders behind the kernels queued on the current stream)
                    dist.all_reduce(host, group=group)
                    view.copy_(host)
                return 0
            except Exception:   # noqa: BLE001 -- must not unwind through the C caller
                import traceback
                traceback.print_exc()
                return 1
        fn = _lib.COLLECTIVE_FN(allreduce)
        _lib.check(self.lib.pmbrl_plan_set_collective(self.plan, fn, None), 'pmbrl_plan_set_collective')
        self._coll = (group, fn)      # keeps the callback alive as long as the plan may call it

    def any_rank(self, flag):
        """True on every rank if `flag` is set on any rank of the group attached with attach_collective (collective:
        every rank calls it).  Lets the ranks that share moment-matching groups take the same decision -- e.g. re-run a
        rollout on a wider-range arithmetic after ONE of them left fp16's range."""
        assert self._coll is not None, 'attach_collective() first'
        group = self._coll[0]
        v = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=self.device)
        if callable(group):
            group(v)
        else:
            import torch.distributed as dist
            if dist.get_backend(group) == 'nccl':
                dist.all_reduce(v, group=group)
            else:
                host = v.cpu()
                dist.all_reduce(host, group=group)
                v = host
        return float(v[0]) > 0.0

    # ------------------------------------------------------------------
    def forward(self, x0, pol_flat, dyn_flat, mx, iSx, my, Sy, pol_scale, pol_bias,
                pol_mask_bits, dyn_mask_bits, z_pol, z_dyn, z_mm=None, z_rr=None, out=None, z_pi=None, u_cat=None):
        """z_pol / z_dyn: [B,.] (frozen, the same at every step) or [H,B,.] (a fresh
        draw per step).  out: optional (states, actions, rewards) tensors to fill."""
        dev = self.device
        t = [_f32c(v, dev) for v in (x0, pol_flat, dyn_flat, mx, iSx, my, Sy, pol_scale,
                                     pol_bias, z_pol, z_dyn)]
        x0, pol_flat, dyn_flat, mx, iSx, my, Sy, pol_scale, pol_bias, z_pol, z_dyn = t
        assert x0.shape == (self.B, self.D), x0.shape
        assert pol_flat.numel() == self.n_pol_params and dyn_flat.numel() == self.n_dyn_params
        assert mx.numel() == self.n_dyn_in and iSx.numel() == self.n_dyn_in and my.numel() == self.D
        zps = zds = 0
        if z_pol.dim() == 3:
            assert z_pol.shape == (self.H, self.B, self.U)
            zps = self.B * self.U
        else:
            assert z_pol.shape[0] >= self.B and z_pol.shape[1] == self.U
        if z_dyn.dim() == 3:
            assert z_dyn.shape == (self.H, self.B, self.D)
            zds = self.B * self.D
        else:
            assert z_dyn.shape[0] >= self.B and z_dyn.shape[1] == self.D
        assert len(pol_mask_bits) == self.n_pol_layers - 1
        assert len(dyn_mask_bits) == self.n_dyn_layers - 1
        if z_mm is not None:
            z_mm = _f32c(z_mm, dev)
            if z_mm.dim() == 3:   # fresh rows per step (PMBRL_FLAG_ZMM_PER_STEP)
                assert z_mm.shape == (self.H, self.cfg.B_global, self.D)
            else:
                assert z_mm.shape[0] >= self.cfg.B_global
        if z_rr is not None:
            z_rr = _f32c(z_rr, dev)
            if z_rr.dim() == 3:
                assert z_rr.shape == (self.H, self.cfg.B_global, 1)
            else:
                assert z_rr.shape[0] >= self.cfg.B_global
        inp = _lib.Inputs()
        inp.x0, inp.pol_params, inp.dyn_params = x0.data_ptr(), pol_flat.data_ptr(), dyn_flat.data_ptr()
        inp.mx, inp.iSx, inp.my, inp.Sy = mx.data_ptr(), iSx.data_ptr(), my.data_ptr(), Sy.data_ptr()
        inp.pol_scale, inp.pol_bias = pol_scale.data_ptr(), pol_bias.data_ptr()
        # frozen masks: bit rows [>= B, nt]; per-step masks (FLAG_*_MASKS_PER_STEP): [H * B, nt]
        for i, b in enumerate(pol_mask_bits):
            assert b.shape[0] >= self.mask_rows[0] and b.is_contiguous()
            inp.pol_mask_bits[i] = b.data_ptr()
        for i, b in enumerate(dyn_mask_bits):
            assert b.shape[0] >= self.mask_rows[1] and b.is_contiguous()
            inp.dyn_mask_bits[i] = b.data_ptr()
        inp.z_pol, inp.z_dyn = z_pol.data_ptr(), z_dyn.data_ptr()
        inp.z_pol_step_stride, inp.z_dyn_step_stride = zps, zds
        inp.z_mm = z_mm.data_ptr() if z_mm is not None else None
        inp.z_rr = z_rr.data_ptr() if z_rr is not None else None
        if self.n_comp:   # mixture head: frozen Gumbel noise [B, n], uniforms of the component draws [H, B]
            z_pi, u_cat = _f32c(z_pi, dev), _f32c(u_cat, dev)
            assert z_pi.shape == (self.B, self.n_comp) and u_cat.shape == (self.H, self.B)
            assert z_dyn.dim() == 3, 'mixture head: z_dyn is a per-step draw [H, B, D]'
            inp.z_pi, inp.u_cat = z_pi.data_ptr(), u_cat.data_ptr()
        self._inputs = inp
        self._keep = (t, z_mm, z_rr, list(pol_mask_bits), list(dyn_mask_bits), z_pi, u_cat)
        # `out`: the caller's tensors receive this rollout's trajectory and are what backward() reads;
        # they are NOT adopted as the engine's own buffers (engines are shared between callers by
        # shape: a later rollout must not write into tensors an earlier one handed out)
        S, A, R = out if out is not None else (self.states, self.actions, self.rewards)
        assert S.shape == (self.H + 1, self.B, self.D) and S.is_contiguous()
        assert A.shape == (self.H, self.B, self.U) and A.is_contiguous()
        assert R.numel() == self.H * self.B and R.is_contiguous()
        self._traj = (S, A, R)
        self.generation += 1
        _lib.check(self.lib.pmbrl_rollout_fwd(self.plan, _stream(), self._ws_ptr, C.byref(inp),
                                              _ptr(S), _ptr(A), _ptr(R), _ptr(self.status)),
                   'pmbrl_rollout_fwd')
        return S, A, R

    def valid_steps(self):
        """Host sync: number of steps completed before a numerical failure (H if none)."""
        s = int(self.status[0].item())
        return self.H if s >= self.H else s

    def sweep_failed(self):
        """Host sync: did the last adjoint sweep report a failure of its own (device-wide barrier of a
        moment-matching group spanning workgroups timed out)?"""
        return int(self.status[1].item()) != 0

    def set_loss(self, weights, out=None):
        """dL/dr weights [H, B] kept resident: every forward() then also leaves sum_{t < valid, b} w r in the returned
        one-element tensor (pmbrl_plan_set_loss: the reduction is queued by the forward call).  None switches it off."""
        if weights is None:
            _lib.check(self.lib.pmbrl_plan_set_loss(self.plan, None, None), 'pmbrl_plan_set_loss')
            self._loss = None
            return None
        w = _f32c(weights.reshape(self.H, self.B), self.device)
        out = out if out is not None else torch.zeros(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pmbrl_plan_set_loss(self.plan, _ptr(w), _ptr(out)), 'pmbrl_plan_set_loss')
        self._loss = (w, out)      # keeps both alive as long as the plan may use them
        return out

    def backward(self, grad_rewards, grad_states=None, grad_actions=None, want_x0=False,
                 want_agn=False, adam=None):
        assert self._inputs is not None, 'forward() first'
        dev = self.device
        gr = _f32c(grad_rewards.reshape(self.H, self.B), dev)
        gs = _f32c(grad_states, dev) if grad_states is not None else None
        ga = _f32c(grad_actions, dev) if grad_actions is not None else None
        if gs is not None:
            assert gs.shape == (self.H + 1, self.B, self.D)
        if ga is not None:
            assert ga.shape == (self.H, self.B, self.U)
        gx0 = torch.empty((self.B, self.D), dtype=torch.float32, device=dev) if want_x0 else None
        agn = torch.empty((self.H, self.B), dtype=torch.float32, device=dev) if want_agn else None
        S, A, R = self._traj
        # the status word makes the adjoint cover exactly the steps the forward sweep completed
        # (truncated horizon, utils/rollout.py:154-157), decided on the device
        if adam is not None:
            # the optimiser step in the same call (pmbrl_rollout_bwd_adam): adam = dict(params, exp_avg, exp_avg_sq,
            # step (device int64), lr, betas, eps, max_norm, [norm_out], [expect], [loss_out]); loss_out: a one-element
            # float32 tensor that receives sum(grad_rewards * rewards) over the valid steps from this call's own
            # launches (pmbrl_adam::loss_out_d) -- no set_loss / weighted_sum launch needed then
            for k in ('params', 'exp_avg', 'exp_avg_sq'):
                t = adam[k]
                assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.n_pol_params
            assert adam['step'].is_cuda and adam['step'].dtype == torch.int64
            o = _lib.Adam(adam['params'].data_ptr(), adam['exp_avg'].data_ptr(), adam['exp_avg_sq'].data_ptr(),
                          adam['step'].data_ptr(), float(adam['lr']), float(adam['betas'][0]), float(adam['betas'][1]),
                          float(adam['eps']), float(adam.get('max_norm') or 0.0),
                          adam['norm_out'].data_ptr() if adam.get('norm_out') is not None else None,
                          int(adam.get('expect') or 0),
                          adam['loss_out'].data_ptr() if adam.get('loss_out') is not None else None)
            if adam.get('loss_out') is not None:
                lo = adam['loss_out']
                assert lo.is_cuda and lo.dtype == torch.float32 and lo.numel() >= 1
            _lib.check(self.lib.pmbrl_rollout_bwd_adam(self.plan, _stream(), self._ws_ptr,
                                                       C.byref(self._inputs), _ptr(S), _ptr(A), _ptr(R), _ptr(gr),
                                                       _ptr(gs), _ptr(ga), _ptr(self.grad_flat), _ptr(gx0),
                                                       _ptr(agn), _ptr(self.status), C.byref(o)),
                       'pmbrl_rollout_bwd_adam')
            return self.grad_flat, gx0, agn
        _lib.check(self.lib.pmbrl_rollout_bwd(self.plan, _stream(), self._ws_ptr,
                                              C.byref(self._inputs), _ptr(S), _ptr(A), _ptr(R), _ptr(gr),
                                              _ptr(gs), _ptr(ga), _ptr(self.grad_flat), _ptr(gx0),
                                              _ptr(agn), _ptr(self.status)),
                   'pmbrl_rollout_bwd')
        return self.grad_flat, gx0, agn

    def set_replay(self, on=1):
        """0: never replay repeated calls as hipGraphs; 1: the library's default (the one-launch-per-step forms); 2: every
        form (pmbrl_plan_set_replay)."""
        _lib.check(self.lib.pmbrl_plan_set_replay(self.plan, int(on)), 'set_replay')
        info = (C.c_int32 * _lib.INFO_COUNT)()
        _lib.check(self.lib.pmbrl_plan_info(self.plan, info), 'plan_info')
        self.info['replay'] = int(info[17])

    def reg_calls(self):
        """(forward sweeps, adjoint sweeps) of this plan that ran on the register-resident family (csrc/pmbrl_reg.h)."""
        info = (C.c_int32 * _lib.INFO_COUNT)()
        _lib.check(self.lib.pmbrl_plan_info(self.plan, info), 'plan_info')
        return int(info[19]), int(info[20])

    def replay_count(self):
        """(forward calls, adjoint calls) that went out as one graph launch so far."""
        n = (C.c_int64 * 2)()
        _lib.check(self.lib.pmbrl_plan_replay_count(self.plan, n), 'replay_count')
        return int(n[0]), int(n[1])

    def set_timing(self, on=True):
        _lib.check(self.lib.pmbrl_plan_set_timing(self.plan, 1 if on else 0), 'set_timing')

    def read_timing(self):
        """dict kernel-name -> ms of the last fwd/bwd calls (syncs on the events)."""
        ms = (C.c_float * _lib.TIMER_COUNT)()
        _lib.check(self.lib.pmbrl_plan_read_timing(self.plan, ms), 'read_timing')
        return {n: float(ms[i]) for i, n in enumerate(_lib.TIMER_NAMES)}

    # ------------------------------------------------------------------
    def weighted_sum(self, a, w, out=None):
        """sum_{t < n, b} a[t, b] w[t, b] over the n valid steps of the last forward sweep (the status
        word is read on the device); a, w: [H, B(,1)]."""
        a = _f32c(a.reshape(-1), self.device)
        w = _f32c(w.reshape(-1), self.device)
        assert a.numel() == self.H * self.B and w.numel() == a.numel()
        if out is None:
            out = torch.empty(1, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.pmbrl_weighted_sum_steps(_stream(), _ptr(a), _ptr(w), self.B, self.H,
                                                     _ptr(self.status), _ptr(out)),
                   'pmbrl_weighted_sum_steps')
        return out


def clip_adam(params, grads, exp_avg, exp_avg_sq, step, lr, betas=(0.9, 0.999), eps=1e-8,
              max_norm=None, norm_out=None):
    """Fused clip_grad_norm_ + Adam.step on flat fp32 buffers (pmbrl_clip_adam)."""
    lib = _lib.load()
    for t in (params, grads, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    n = params.numel()
    _lib.check(lib.pmbrl_clip_adam(_stream(), _ptr(params), _ptr(grads), _ptr(exp_avg),
                                   _ptr(exp_avg_sq), n, int(step), float(lr), float(betas[0]),
                                   float(betas[1]), float(eps),
                                   float(max_norm) if max_norm else 0.0, _ptr(norm_out)),
               'pmbrl_clip_adam')


class Graph:
    """hipGraph of library calls (pmbrl_graph_capture_* / pmbrl_graph_launch): `fn()` is recorded once on a side stream
    and replayed with one launch.  fn may call Engine.forward (with out=...), weighted_sum (with out=...), backward,
    clip_adam_guarded, BnnStep ... -- anything that only QUEUES work on the current stream and allocates nothing (pass the
    output tensors in).  Where it pays: the sweeps that are one launch per step (moment-matching groups beyond a
    workgroup, wide states), 200-400 launches per iteration.  NOT recordable: collectives that go through the host
    (Engine.attach_collective with a torch.distributed group) and the peer-to-peer all-reduce of distributed.P2P (its
    generation counter is a kernel argument; the library call returns an error on a recording stream)."""

    def __init__(self, fn, warmup=1):
        self.lib = _lib.load()
        self.handle = C.c_void_p()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):       # (first calls may create events / set attributes: not inside a capture)
                fn()
            side.synchronize()
            _lib.check(self.lib.pmbrl_graph_capture_begin(C.c_void_p(side.cuda_stream)), 'pmbrl_graph_capture_begin')
            try:
                fn()
            finally:
                rc = self.lib.pmbrl_graph_capture_end(C.c_void_p(side.cuda_stream), C.byref(self.handle))
            _lib.check(rc, 'pmbrl_graph_capture_end')
        torch.cuda.current_stream().wait_stream(side)
        self._keep = fn       # the closure keeps the captured tensors alive

    def replay(self):
        _lib.check(self.lib.pmbrl_graph_launch(self.handle, _stream()), 'pmbrl_graph_launch')

    def num_nodes(self):
        n = C.c_int64(0)
        _lib.check(self.lib.pmbrl_graph_num_nodes(self.handle, C.byref(n)), 'pmbrl_graph_num_nodes')
        return int(n.value)

    def __del__(self):
        try:
            if self.handle:
                self.lib.pmbrl_graph_destroy(self.handle)
                self.handle = C.c_void_p()
        except Exception:
            pass


def clip_adam_guarded(params, grads, exp_avg, exp_avg_sq, step_dev, lr, status, expect, betas=(0.9, 0.999),
                      eps=1e-8, max_norm=None, norm_out=None):
    """clip + Adam taken on the device only if `status` (the rollout's status word) says all
    `expect` horizon steps completed; `step_dev` is the device-side int64 step counter
    (pmbrl_clip_adam_guarded)."""
    lib = _lib.load()
    for t in (params, grads, exp_avg, exp_avg_sq):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()
    assert step_dev.is_cuda and step_dev.dtype == torch.int64 and status.is_cuda and status.dtype == torch.int32
    _lib.check(lib.pmbrl_clip_adam_guarded(_stream(), _ptr(params), _ptr(grads), _ptr(exp_avg),
                                           _ptr(exp_avg_sq), params.numel(), _ptr(step_dev), float(lr),
                                           float(betas[0]), float(betas[1]), float(eps),
                                           float(max_norm) if max_norm else 0.0, _ptr(norm_out),
                                           _ptr(status), int(expect)), 'pmbrl_clip_adam_guarded')


def debug_linear(x, W, b, transpose_w=False):
    """Test hook: y = x W^T + b through the kernels' MFMA tile routine."""
    lib = _lib.load()
    R, K = x.shape
    O = W.shape[1] if transpose_w else W.shape[0]
    y = torch.empty((R, O), dtype=torch.float32, device=x.device)
    n_kb, n_ot = (K + 15) // 16, (O + 15) // 16
    scratch = torch.empty(n_kb * n_ot * 256 + n_ot * 16 + 64, dtype=torch.float32, device=x.device)
    _lib.check(lib.pmbrl_debug_linear(_stream(), _ptr(x.contiguous()), _ptr(W.contiguous()),
                                      _ptr(b.contiguous()), R, K, O, 1 if transpose_w else 0,
                                      _ptr(y), _ptr(scratch)), 'pmbrl_debug_linear')
    return y
