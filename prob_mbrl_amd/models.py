"""Host-side mirror of the reference's `prob_mbrl.models` module API for the
MC-PILCO path (reference: models/core.py, models/modules.py, models/densities.py).

These classes are STORAGE + bookkeeping: torch Parameters / buffers with the
reference's names (`model.fc0.weight`, `model.drop0.noise`, `mx`, `Sy`, `scale`
...), so reference checkpoints load and `resample()` / `set_dataset()` /
`regularization_loss()` behave the same.  The arithmetic of the hot path
(Linear -> ReLU -> dropout ... -> density sample -> squash / reward) is NOT done
here: `prob_mbrl_amd.rollout.rollout` hands the tensors to the fused HIP kernels.
"""
import copy
import inspect
import math
from collections import OrderedDict
from collections.abc import Iterable
from functools import partial

import numpy as np
import torch
from torch import nn


class StochasticModule(nn.Module):
    """Marker base (models/modules.py:9-11)."""


# ---------------------------------------------------------------------------
# dropout layers: persistent (PEGASUS) masks
# ---------------------------------------------------------------------------
class BDropout(StochasticModule):
    """Bernoulli dropout with a persistent mask (models/modules.py:14-70).
    The masked activation is divided by the keep probability p."""

    def __init__(self, rate=0.5, name=None, regularizer_scale=1.0, **kwargs):
        super().__init__(**kwargs)
        self.name = name
        rate = rate if isinstance(rate, torch.Tensor) else torch.tensor(rate)
        self.register_buffer('regularizer_scale', torch.tensor(0.5 * regularizer_scale))
        self.register_buffer('rate', rate)
        self.register_buffer('p', 1 - self.rate)
        self.register_buffer('noise', torch.bernoulli(self.p))

    # L2 terms of Gal & Ghahramani (models/modules.py:30-35)
    def weights_regularizer(self, weights):
        self.p = 1 - self.rate
        return self.regularizer_scale * (self.p * (weights**2).sum(0)).sum()

    def biases_regularizer(self, biases):
        return self.regularizer_scale * ((biases**2).sum(0)).sum()

    def resample(self, seed=None):
        self.update_noise(self.noise, seed)

    def update_noise(self, x, seed=None):
        """Redraw the mask with the shape of x (models/modules.py:40-44); re-seeds the
        GLOBAL torch RNG when a seed is given, like the reference."""
        if seed is not None:
            torch.manual_seed(int(seed))
        self.p = 1 - self.rate
        self.noise.data = torch.bernoulli(self.p.expand(x.shape).to(self.noise.device))
        # `.data =` keeps the tensor's version counter and the allocator may hand back a previously
        # used address: the packed-bit cache of the rollout keys on this counter as well
        self._mask_gen = getattr(self, '_mask_gen', 0) + 1

    # --- what the fused rollout needs -------------------------------------
    def keep_prob(self):
        # memoised: float() of a device tensor is a host-device sync
        sig = (self.rate.data_ptr(), self.rate._version, str(self.rate.device))
        hit = getattr(self, '_keep_cache', None)
        if hit is not None and hit[0] == sig:
            return hit[1]
        p = (1 - self.rate)
        if p.numel() != 1:
            # per-unit rates (models/modules.py:19-27 takes a tensor): the kernels scale a layer by ONE 1 / p, so
            # the rollout folds 1 / p_j into row j of the layer that feeds this dropout (unit_inv_keep below;
            # m_j relu(v_j) / p_j = m_j relu(v_j / p_j): ReLU is positively homogeneous) and runs the layer with 1
            self._keep_cache = (sig, 1.0)
            return 1.0
        self._keep_cache = (sig, float(p))
        return self._keep_cache[1]

    def unit_inv_keep(self):
        """1 / p per unit [width] when the rate is a tensor (see keep_prob), else None."""
        if type(self) is not BDropout or self.rate.numel() == 1:
            return None
        return (1.0 / (1 - self.rate)).detach().reshape(-1).float()

    def hard_mask(self, B, width):
        """{0,1} mask [>=B, width]; (re)drawn when the stored one cannot be reused, with the
        reference's rule (models/modules.py:48-54): wrong trailing shape or too few rows."""
        n = self.noise
        if n.dim() != 2 or n.shape[1] != width or n.shape[0] < B:
            self.update_noise(torch.empty(B, width))
        return self.noise

    def forward_mask(self, B, width, resample=True, seed=None):
        """The {0,1} mask a stand-alone forward over B rows applies, with the reference's
        rules (models/modules.py:46-61): redraw-and-store when the stored mask cannot be
        reused, a fresh unstored draw when `resample`, else the stored rows."""
        n = self.noise
        if n.dim() != 2 or n.shape[1] != width or B > n.shape[0]:
            self.update_noise(torch.empty(B, width), seed)
        elif resample:
            if seed is not None:
                torch.manual_seed(int(seed))
            self.p = 1 - self.rate
            return torch.bernoulli(self.p.expand(B, width).to(self.noise.device))
        return self.noise[:B]

    def step_mask_bits(self, H, B, width):
        """Bit rows [H * B, ceil(width/16)] of H fresh masks -- what a rollout with resample=True at every step
        applies (models/modules.py:55-58: drawn, not stored) -- from ONE device launch (pmbrl_draw_masks) instead of
        H x (rand, compare, pack) eager launches.  The counter-based generator is keyed from torch's CPU generator,
        so torch.manual_seed makes the draw reproducible; it is not the draw torch.bernoulli would have made."""
        from . import engine as E
        seed = int(torch.randint(0, 2**62, (1,)))
        bits, _ = E.draw_masks('bernoulli', seed, 0, (1 - self.rate).to(self.noise.device), 0.0, H * B, width)
        return bits

    def forward(self, x, resample=True, mask_dims=2, seed=None, **kwargs):
        raise NotImplementedError(
            'a dropout layer is not evaluated on its own on the device path; call the network '
            '(Policy / Regressor / DynamicsModel) or prob_mbrl_amd.utils.rollout')

    def extra_repr(self):
        return 'rate={}, regularizer_scale={}'.format(self.rate, self.regularizer_scale)


class CDropout(BDropout):
    """Concrete dropout (models/modules.py:73-171).  In eval mode (how mc_pilco runs the
    dynamics model, algorithms/mc_pilco.py:43) the mask is a hard Bernoulli draw from the
    concrete probabilities and is NOT divided by p."""

    def __init__(self, rate=0.5, name=None, regularizer_scale=1.0, dropout_regularizer=1.0,
                 temperature=0.1, **kwargs):
        if not isinstance(rate, torch.Tensor):
            rate = torch.tensor(np.asarray(rate), dtype=torch.get_default_dtype())
        super().__init__(rate, name, regularizer_scale, **kwargs)
        self.register_buffer('temp', torch.tensor(temperature))
        self.register_buffer('dropout_regularizer', torch.tensor(dropout_regularizer))
        self.logit_p = nn.Parameter(-torch.log(1.0 / self.p - 1.0))
        self.register_buffer('concrete_noise', torch.bernoulli(self.p))

    def weights_regularizer(self, weights):
        p = self.p
        reg = self.regularizer_scale * (p * (weights**2).sum(0))
        reg = reg + self.dropout_regularizer * (p * p.log() + (1 - p) * (1 - p).log())
        return reg.sum()

    def update_noise(self, x, seed=None):
        if seed is not None:
            torch.manual_seed(int(seed))
        self.noise.data = torch.rand(x.shape, device=self.noise.device, dtype=self.noise.dtype)
        self._mask_gen = getattr(self, '_mask_gen', 0) + 1
        if not self.training:
            self.update_concrete_noise(self.noise)

    def update_concrete_noise(self, noise):
        """models/modules.py:102-118: probs = sigmoid((logit_p + log((u+1e-7)/(1-(u-1e-7))))/temp),
        hard sample with a straight-through value (exactly {0,1})."""
        concrete_p = self.logit_p + ((noise + 1e-7) / (1 - (noise - 1e-7))).log()
        probs = (concrete_p / self.temp).sigmoid()
        forced = getattr(self, '_forced_sample', None)      # tests: replay recorded draws
        hard = torch.bernoulli(probs) if forced is None else forced.to(probs)
        self.concrete_noise = (hard - probs).detach() + probs
        self._mask_gen = getattr(self, '_mask_gen', 0) + 1
        self.p = self.logit_p.sigmoid()

    def keep_prob(self):
        return 1.0   # x * concrete_noise, no division (models/modules.py:158-160)

    def forward_mask(self, B, width, resample=False, seed=None):
        """Mask of a stand-alone forward (models/modules.py:120-160).  Eval mode: the stored hard
        sample (redrawn only when the noise is).  Training mode: a fresh Bernoulli sample of the
        relaxed probabilities at every call, like the reference -- its VALUE only: a stand-alone
        forward is not differentiable with respect to the network here (the training step is
        pmbrl_bnn_loss_grad)."""
        resampled = False
        noise = self.noise
        c = self.concrete_noise
        if resample:
            if seed is not None:
                torch.manual_seed(int(seed))
            noise = torch.rand(B, width, device=self.noise.device, dtype=self.noise.dtype)
            resampled = True
        elif (noise.dim() != 2 or c.dim() != 2 or noise.shape[1] != width or c.shape[1] != width
              or B > c.shape[0]):
            self.update_noise(torch.empty(B, width), seed)
            noise = self.noise
            resampled = True
        if self.training or resampled:
            self.update_concrete_noise(noise)
        return self.concrete_noise.detach()[:B]

    def step_mask_bits(self, H, B, width):
        """H fresh eval-mode masks as bit rows, one device launch (see BDropout.step_mask_bits): new uniform noise and
        a hard sample of its concrete probabilities per step (models/modules.py:134-139,155-157).  Like the reference,
        the module is left holding the LAST step's hard sample (concrete_noise) while its stored uniform noise stays
        untouched: forward(resample=True) draws into a local (models/modules.py:139-143)."""
        if self.training:
            raise NotImplementedError('training-mode (relaxed) concrete dropout is not on the rollout path; call '
                                      'dynamics.eval() like mc_pilco does')
        from . import engine as E
        seed = int(torch.randint(0, 2**62, (1,)))
        lp = self.logit_p.detach()
        if lp.numel() not in (1, width):
            raise ValueError('logit_p has %d entries for a layer of width %d' % (lp.numel(), width))
        bits, aux = E.draw_masks('concrete', seed, 0, lp, float(self.temp), H * B, width, aux=((H - 1) * B, B))
        self.concrete_noise = aux['hard']
        self._mask_gen = getattr(self, '_mask_gen', 0) + 1
        self.p = self.logit_p.sigmoid()
        return bits

    def hard_mask(self, B, width):
        if self.training:
            raise NotImplementedError('training-mode (relaxed) concrete dropout is not on the '
                                      'rollout path; call dynamics.eval() like mc_pilco does')
        n, c = self.noise, self.concrete_noise
        if (n.dim() != 2 or c.dim() != 2 or n.shape[1] != width or c.shape[1] != width
                or c.shape[0] < B):
            self.update_noise(torch.empty(B, width))
        return self.concrete_noise.detach()


# ---------------------------------------------------------------------------
# output density
# ---------------------------------------------------------------------------
class DiagGaussianDensity(StochasticModule):
    """Diagonal Gaussian head (models/densities.py:70-148): the network emits
    [mean | log_std]; log_std is soft-clamped at log(max_noise_std); samples use the
    persistent noise buffer z."""

    def __init__(self, output_dims, max_noise_std=5.0):
        super().__init__()
        self.output_dims = output_dims
        self.register_buffer('z', torch.ones([1, 1]))
        self.register_buffer('max_log_std', torch.tensor(max_noise_std).log())
        self.expl_scale = 1.0

    def resample(self, seed=None):
        if seed is not None:
            torch.manual_seed(int(seed))
        self.z.data = torch.randn_like(self.z)

    def frozen_noise(self, B, resample_noise, seed=None):
        """z [B, dims] with the reference's refresh rule (models/densities.py:113-116)."""
        D = int(self.output_dims)
        if tuple(self.z.shape) != (B, D) or resample_noise:
            if seed is not None:
                torch.manual_seed(int(seed))
            self.z.data = torch.randn(B, D, device=self.z.device, dtype=self.z.dtype)
        return self.z

    def log_prob(self, z, mean, log_std=None):
        """models/densities.py:123-144 (used by BNN training, not by the rollout)."""
        deltas = mean - z
        if log_std is None:
            return -(deltas**2).sum(-1) * 0.5
        return (-0.5 * ((deltas * (-log_std).exp())**2).sum(-1) - log_std.sum(-1) -
                self.output_dims * 0.5 * math.log(2 * math.pi))

    def forward(self, x, **kwargs):
        raise NotImplementedError('stand-alone density forward is not part of the accelerated path')

    def __repr__(self):
        return self.__class__.__name__ + '(output_dims=%d)' % self.output_dims


class GaussianMixtureDensity(StochasticModule):
    """Mixture-of-diagonal-Gaussians head (models/densities.py:151-259; examples/deep_pilco_mm.py:117-121): the
    network emits [n D means | n D log-stds | n component logits | 1 log-temperature] (width (2 D + 1) n + 1); a
    sample picks ONE component -- straight-through one-hot over the tempered Gumbel-softmax of the logits -- and
    adds that component's Gaussian noise.  Offered as the dynamics model's `output_density` inside rollouts (the
    sampling phase of the general kernel family and its adjoint: csrc/pmbrl_rollout.h); `log_prob` is the mixture
    log-likelihood (torch ops, for callers that evaluate it themselves).

    Random draws, as the reference makes them: `z_pi` (Gumbel noise, [B, n]) is frozen until `resample()` / a
    shape change / resample_noise=True; the Gaussian noise is redrawn at EVERY step whatever resample_noise says
    (the reference compares `mean[:-1].shape` with `z_pi.shape`, :228-231) and so is the component index
    (`Categorical(k_soft).sample()`, :221-222).  On the device both per-step draws are explicit inputs of the
    rollout: `z_normal` [H, B, D] and uniforms [H, B] for an inverse-CDF draw."""

    def __init__(self, output_dims, n_components, max_noise_std=5.0):
        super().__init__()
        self.n_components = int(n_components)
        self.output_dims = output_dims
        self.register_buffer('z_normal', torch.ones([1, 1]))
        self.register_buffer('z_pi', torch.ones([1, 1]))
        self.register_buffer('max_log_std', torch.tensor(max_noise_std).log())
        # True: differentiate the noise term with each step's own noise; False (default) reproduces the
        # reference's gradient, whose autograd ends up using the LAST step's noise at every step (see
        # oracle/ref_torch.py:gmm_sample)
        self.exact_noise_grad = False

    def resample(self, seed=None):
        if seed is not None:
            torch.manual_seed(int(seed))
        u = torch.rand_like(self.z_pi)
        self.z_pi.data = -(-u.log()).log()
        self.z_normal.data = torch.randn_like(self.z_normal)

    def frozen_gumbel(self, B, resample_noise):
        """z_pi [B, n] with the reference's refresh rule (models/densities.py:213-216)."""
        n = self.n_components
        if tuple(self.z_pi.shape) != (B, n) or resample_noise:
            u = torch.rand(B, n, device=self.z_pi.device, dtype=self.z_pi.dtype)
            self.z_pi.data = -(-u.log()).log()
        return self.z_pi

    def from_head(self, x, my=None, Sy=None, return_samples=False, resample_noise=True, sampling_temperature=0.1):
        """models/densities.py:173-233 on the raw head rows x [B, (2 D + 1) n + 1]: (mean, log_std, logit_pi) with
        mean / log_std [B, D, n], or samples [B, D]."""
        D, n = int(self.output_dims), self.n_components
        nD = D * n
        mean, log_std, extras = x[:, :nD], x[:, nD:2 * nD], x[:, 2 * nD:]
        logit_pi, log_temperature = extras[:, :n], extras[:, n:]
        log_std = -torch.nn.functional.softplus(-log_std + self.max_log_std) + self.max_log_std
        mean, log_std = mean.reshape(-1, D, n), log_std.reshape(-1, D, n)
        logit_pi = logit_pi / (1e-1 + torch.nn.functional.softplus(log_temperature))
        if my is not None and Sy is not None:
            log_std = log_std + Sy.reshape(1, D, 1).log()
            mean = mean * Sy.reshape(1, D, 1) + my.reshape(1, D, 1)
        if not return_samples:
            return mean, log_std, logit_pi
        z1 = self.frozen_gumbel(x.shape[0], resample_noise)
        k_soft = ((torch.log_softmax(logit_pi, -1) + z1) / sampling_temperature).softmax(-1)
        k_idx = torch.distributions.Categorical(k_soft).sample().view(-1, 1)
        k = torch.zeros_like(k_soft).scatter(1, k_idx, 1)[:, None, :]
        self.z_normal.data = torch.randn(x.shape[0], D, device=x.device, dtype=x.dtype)   # (:228-231: every call)
        return (mean * k).sum(-1) + self.z_normal * (log_std * k).sum(-1).exp()

    def log_prob(self, z, mean, log_std, logit_pi):
        """models/densities.py:235-252: log sum_c pi_c N(z; mean_c, diag(std_c^2)), mean / log_std [B, D, n]."""
        D = int(self.output_dims)
        deltas = mean - z.unsqueeze(-1)
        log_norm = -D * 0.5 * math.log(2 * math.pi) - log_std.sum(-2)
        dists = -0.5 * ((deltas * (-log_std).exp())**2).sum(-2)
        log_probs = torch.log_softmax(logit_pi, -1) + log_norm + dists
        return torch.logsumexp(log_probs, dim=-1, keepdim=True)

    def forward(self, x, **kwargs):
        raise NotImplementedError('stand-alone density forward is not part of the accelerated path')

    def __repr__(self):
        return self.__class__.__name__ + '(output_dims=%d, n_components=%d)' % (self.output_dims, self.n_components)


# ---------------------------------------------------------------------------
# containers
# ---------------------------------------------------------------------------
class BSequential(nn.Sequential):
    """Sequential with resampling control and the dropout regulariser
    (models/modules.py:198-274)."""

    def __init__(self, *args):
        super().__init__(*args)
        self.modules_to_regularize = []

    def resample(self, seed=None):
        i = 0
        for module in self._modules.values():
            if isinstance(module, BDropout):
                module.resample(seed + i if seed is not None else None)
                i += 1

    def regularization_loss(self):
        """For every dropout layer, regularise the next Linear's weights/biases."""
        mods = list(self._modules.values())
        total = 0
        for i, m in enumerate(mods):
            if hasattr(m, 'weights_regularizer'):
                for nxt in mods[i:]:
                    if isinstance(nxt, nn.Linear):
                        total = total + m.weights_regularizer(nxt.weight)
                        if nxt.bias is not None and hasattr(m, 'biases_regularizer'):
                            total = total + m.biases_regularizer(nxt.bias)
                        break
            elif hasattr(m, 'regularization_loss'):
                total = total + m.regularization_loss()
        return total

    def forward(self, input, **kwargs):
        raise NotImplementedError('call the owning Policy / Regressor / DynamicsModel: the network is '
                                  'evaluated by one device kernel together with its normalisation and head')

    def device_forward(self, x, density, resample=True, seed=None, return_samples=False,
                       resample_noise=True, in_shift=None, in_iscale=None, out_scale=None,
                       out_shift=None, squash=None, **kwargs):
        """Evaluate the network + Gaussian head on the rows of x with pmbrl_mlp_forward
        (models/modules.py:215-232 + models/densities.py:87-121).  Returns samples [B, n_out]
        (squashed if `squash=(scale, bias)`) or (mean, log_std)."""
        from . import engine as E
        from .rollout import flat_parameters
        linears, drops, inner = self.layer_spec()
        density = inner if inner is not None else density
        if not x.is_cuda:
            raise RuntimeError('the network lives on a HIP device: pass a device tensor (no CPU fallback)')
        x = x.to(torch.float32)
        B = x.shape[0]
        dims = [linears[0].in_features] + [l.out_features for l in linears]
        flat, _ = flat_parameters(linears, self)
        # per-unit BDropout rates: keep_prob() is 1 for them and 1 / p_j lives in row j of the layer in front of the
        # dropout, exactly as the rollout sees the network (rollout.Bundle.scale_unit_rows; models/modules.py:55-61)
        from .rollout import Bundle
        unit = Bundle._unit_rows(drops, dims, x.device)
        if unit:
            flat = Bundle.scale_unit_rows(flat.detach(), unit)
        mixture = isinstance(density, GaussianMixtureDensity)
        plain = density is None or mixture
        if plain:
            # a network without an output density (the critic of
            # examples/deep_pilco_no_mm_with_value.py:269-278; models/core.py:185-186:
            # out * Sy + my).  The kernel always evaluates a Gaussian head, so give it one whose
            # log-std rows are zero and read back the (scaled) mean.
            O, K = dims[-1], dims[-2]
            n_last = O * K + O
            flat = torch.cat([flat[:flat.numel() - n_last], flat[-n_last:-O], flat.new_zeros(O * K),
                              flat[-O:], flat.new_zeros(O)])
            dims = dims[:-1] + [2 * O]
        keep, bits = [], []
        for lin, dr in zip(linears[:-1], drops):
            if dr is None:
                keep.append(1.0)
                bits.append(None)
            else:
                m = dr.forward_mask(B, lin.out_features, resample=resample, seed=seed)
                keep.append(dr.keep_prob())
                bits.append(E.pack_mask(m.to(device=x.device, dtype=torch.float32)))
        if mixture:
            # the network on the device (raw head rows), the few head formulas of the mixture with torch ops on
            # those rows (stand-alone evaluation is not the hot path; rollouts and training are fused kernels)
            o = E.mlp_forward(x, flat, dims, keep, bits, None, in_shift, in_iscale, None, None, want=('mean',))['mean']
            return density.from_head(o, out_shift, out_scale, return_samples=return_samples,
                                     resample_noise=resample_noise)
        if plain:
            return E.mlp_forward(x, flat, dims, keep, bits, None, in_shift, in_iscale, out_scale,
                                 out_shift, want=('mean',))['mean']
        z = None
        if return_samples:
            z = density.frozen_noise(B, resample_noise, seed).to(device=x.device, dtype=torch.float32)
        out = E.mlp_forward(x, flat, dims, keep, bits, z, in_shift, in_iscale, out_scale, out_shift,
                            squash[0] if (squash and return_samples) else None,
                            squash[1] if (squash and return_samples) else None,
                            max_log_std=float(density.max_log_std),
                            want=('sample',) if return_samples else ('mean', 'log_std'))
        if return_samples:
            return out['sample']
        return out['mean'], out['log_std']

    # --- structure the fused kernels understand ---------------------------
    def layer_spec(self):
        """Parse Linear -> [ReLU] -> [dropout] ... -> Linear [-> density] into
        (linears, dropouts-per-hidden-layer, density)."""
        linears, drops, density = [], [], None
        pending = None
        for name, m in self._modules.items():
            if isinstance(m, nn.Linear):
                if pending is not None:
                    drops.append(pending['drop'])
                    if not pending['relu']:
                        raise NotImplementedError('hidden layers must use ReLU on the device path')
                pending = dict(drop=None, relu=False)
                linears.append(m)
            elif isinstance(m, nn.ReLU):
                pending['relu'] = True
            elif isinstance(m, BDropout):
                if pending is None:
                    raise NotImplementedError('input dropout is not offered on the device path')
                pending['drop'] = m
            elif isinstance(m, DiagGaussianDensity):
                density = m
            else:
                raise NotImplementedError('module %s (%s) is not offered on the device path' %
                                          (name, type(m).__name__))
        if pending is not None and (pending['drop'] is not None or pending['relu']):
            raise NotImplementedError('the output layer must be a plain Linear')
        return linears, drops, density


def mlp(input_dims, output_dims, hidden_dims=[200, 200], nonlin=nn.ReLU, output_nonlin=None,
        weights_initializer=partial(nn.init.xavier_normal_, gain=nn.init.calculate_gain('relu')),
        biases_initializer=partial(nn.init.uniform_, a=-1e-1, b=1e-1), hidden_biases=True,
        output_biases=True, dropout_layers=BDropout, input_dropout=None, spectral_norm=False,
        spectral_norm_output=False, layer_norm=False):
    """Same factory signature and module names as models/core.py:15-99
    (fc%d / nonlin%d / drop%d / fc_out / fc_nonlin)."""
    if spectral_norm or spectral_norm_output or layer_norm or input_dropout is not None:
        raise NotImplementedError('spectral_norm / layer_norm / input_dropout are outside the '
                                  'accelerated MC-PILCO path')
    if not (hidden_biases and output_biases):
        raise NotImplementedError('bias-free layers are not offered on the device path')
    hidden_dims = list(hidden_dims)
    dims = [int(input_dims)] + [int(h) for h in hidden_dims]
    if not isinstance(dropout_layers, Iterable):
        dropout_layers = [copy.deepcopy(dropout_layers)] * len(hidden_dims)
    if not isinstance(nonlin, Iterable):
        nonlin = [nonlin] * len(hidden_dims)
    mods = OrderedDict()
    for i, (din, dout) in enumerate(zip(dims[:-1], dims[1:])):
        drop_i = dropout_layers[i]
        if inspect.isclass(drop_i):
            drop_i = drop_i(name='drop%d' % i)
        mods['fc%d' % i] = nn.Linear(din, dout, bias=True)
        if callable(nonlin[i]):
            mods['nonlin%d' % i] = nonlin[i]()
        if drop_i is not None:
            mods['drop%d' % i] = drop_i
    mods['fc_out'] = nn.Linear(dims[-1], int(output_dims), bias=True)
    if callable(output_nonlin):
        mods['fc_nonlin'] = output_nonlin()
    net = BSequential(mods)
    for m in net.modules():
        if isinstance(m, nn.Linear):
            if callable(weights_initializer):
                weights_initializer(m.weight)
            if callable(biases_initializer):
                biases_initializer(m.bias)
    net.float()
    return net


# ---------------------------------------------------------------------------
# wrappers
# ---------------------------------------------------------------------------
class _Loadable(nn.Module):
    def load(self, state_dict):
        """Copy every matching parameter / buffer (models/core.py:154-159,214-219)."""
        own = dict(self.named_parameters())
        own.update(self.named_buffers())
        for k, v in state_dict.items():
            if k in own:
                own[k].data = v.data.clone().to(own[k].device)
        for m in self.modules():      # stored masks may have changed under an unchanged version counter
            if isinstance(m, BDropout):
                m._mask_gen = getattr(m, '_mask_gen', 0) + 1

    def regularization_loss(self):
        return self.model.regularization_loss()


class Regressor(_Loadable):
    """models/core.py:121-187: input/output normalisation around a BNN."""

    def __init__(self, model, output_density=None, angle_dims=[]):
        super().__init__()
        self.model = model
        self.output_density = output_density
        self.register_buffer('angle_dims', torch.tensor(angle_dims).long())
        for name, val in (('X', 1.0), ('Y', 1.0), ('mx', 0.0), ('Sx', 1.0), ('iSx', 1.0),
                          ('my', 0.0), ('Sy', 1.0), ('iSy', 1.0)):
            self.register_buffer(name, torch.full([1, 1], val))

    def set_dataset(self, X, Y, N_ensemble=-1, p=0.5):
        if len(self.angle_dims):
            from .utils import to_complex
            X = to_complex(X, self.angle_dims)
        self.X.data = X
        self.Y.data = Y
        self.mx.data = self.X.mean(0, keepdim=True)
        self.Sx.data = 4.0 * self.X.std(0, keepdim=True)
        self.Sx.data[self.Sx == 0] = 4.0
        self.iSx.data = self.Sx.reciprocal()
        self.my.data = self.Y.mean(0, keepdim=True)
        self.Sy.data = 4.0 * self.Y.std(0, keepdim=True)
        self.Sy.data[self.Sy == 0] = 4.0
        self.iSy.data = self.Sy.reciprocal()

    def resample(self, *args, **kwargs):
        self.model.resample(*args, **kwargs)
        if self.output_density is not None:
            self.output_density.resample(*args, **kwargs)

    def forward(self, x, normalize=True, **kwargs):
        """models/core.py:169-187: (mean, log_std) of the predictive Gaussian, or samples with
        return_samples=True; one device kernel (pmbrl_mlp_forward)."""
        if len(self.angle_dims) > 0:
            from .utils import to_complex
            x = to_complex(x, self.angle_dims)
        kwargs.setdefault('resample', True)      # BSequential.forward's default (modules.py:215)
        return self.model.device_forward(
            x, self.output_density,
            in_shift=self.mx if normalize else None, in_iscale=self.iSx if normalize else None,
            out_scale=self.Sy if normalize else None, out_shift=self.my if normalize else None, **kwargs)


class DynamicsModel(Regressor):
    """models/core.py:251-303.  `reward_func` must be one of prob_mbrl_amd.rewards.*
    (an analytic reward with a `.spec(D)`); a learned reward is not offered (the reference's
    own learned-reward branch raises)."""

    def __init__(self, model, reward_func=None, predict_done=False, **kwargs):
        super().__init__(model, **kwargs)
        self.register_buffer('maxR', torch.ones([1, 1]))
        self.register_buffer('minR', torch.ones([1, 1]))
        # a reference-shaped reward module (env.reward_func of the reference's environments: constants as
        # parameters, envs/cartpole/env.py:27-40) is restated as this build's analytic reward of the same name
        from . import rewards
        known = rewards.from_module(reward_func)
        self.reward_func = known if known is not None else reward_func

    def set_dataset(self, X, Y):
        super().set_dataset(X, Y)
        D = self.Y.shape[-1] - 1
        R = self.Y[..., D]
        self.maxR.data = R.max()
        self.minR.data = R.min()

    def forward(self, inputs, separate_outputs=False, deltas=True, **kwargs):
        """models/core.py:265-303."""
        as_tuple = isinstance(inputs, (tuple, list))
        if as_tuple:
            prev_states, actions = inputs[0], inputs[1]
            inputs = torch.cat([prev_states, actions], -1)
        outs = super().forward(inputs, **kwargs)
        if not kwargs.get('return_samples', False):
            return outs
        if not as_tuple:
            D = outs.shape[-1]
            prev_states, actions = inputs[..., :D], inputs[..., D:]
        if not callable(self.reward_func):
            raise NotImplementedError('a learned reward head is not offered (the reference raises here too)')
        dstates = outs
        rewards = self.reward_func(prev_states + dstates, actions)
        states = dstates if deltas else prev_states + dstates
        if separate_outputs:
            return states, rewards
        return torch.cat([states, rewards], -1)


class Policy(_Loadable):
    """models/core.py:190-248: BNN policy with tanh squashing to [minU, maxU]."""

    def __init__(self, model, maxU=1.0, minU=None, angle_dims=[]):
        super().__init__()
        self.model = model
        self.register_buffer('angle_dims', torch.tensor(angle_dims).long())
        if minU is None:
            minU = -maxU
        scale = 0.5 * (maxU - minU)
        bias = 0.5 * (maxU + minU)
        self.register_buffer('scale', torch.as_tensor(scale).squeeze())
        self.register_buffer('bias', torch.as_tensor(bias).squeeze())

    def resample(self, *args, **kwargs):
        self.model.resample(*args, **kwargs)

    def forward(self, x, **kwargs):
        """models/core.py:221-248: squashed action samples for the rows of x (numpy in ->
        numpy out, a single state -> one row), one device kernel (pmbrl_mlp_forward)."""
        return_numpy = isinstance(x, np.ndarray)
        kwargs['resample'] = kwargs.get('resample', True)
        kwargs['return_samples'] = kwargs.get('return_samples', True)
        x = torch.as_tensor(x, dtype=self.scale.dtype, device=self.scale.device)
        if x.dim() == 1:
            x = x[None, :]
        if len(self.angle_dims) > 0:
            from .utils import to_complex
            x = to_complex(x, self.angle_dims)
        u = self.model.device_forward(x, None, squash=(self.scale, self.bias), **kwargs)
        if isinstance(u, tuple):
            # return_samples=False: the reference adds the two heads before squashing (core.py:238-243)
            u = self.scale * (u[0] + u[1]).tanh() + self.bias
        return u.detach().cpu().numpy() if return_numpy else u
