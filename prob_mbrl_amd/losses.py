"""The log-likelihoods `utils.train_regressor(log_likelihood=...)` takes (losses.py:16-64 of the reference).  On
the device path they are SELECTORS: train_regressor recognises them and runs the matching loss inside the fused
BNN training kernel (pmbrl_bnn_loss_grad, loss_kind 0 / 2); called directly they evaluate the same formulas
with torch ops on whatever device their arguments live on."""
import math

import torch


def gaussian_log_likelihood(targets, means, log_stds=None):
    """losses.py:16-37: diagonal covariance."""
    deltas = means - targets
    if log_stds is None:
        return -(deltas**2).sum(-1) * 0.5
    return (-0.5 * ((deltas * (-log_stds).exp())**2).sum(-1) - log_stds.sum(-1) -
            means.shape[-1] * 0.5 * math.log(2 * math.pi))


def gaussian_mixture_log_likelihood(targets, means, log_stds, logit_pi):
    """losses.py:40-64: means / log_stds [batch, output_dims, n_components]."""
    D = means.shape[-2]
    deltas = means - targets.unsqueeze(-1)
    log_norm = -D * 0.5 * math.log(2 * math.pi) - log_stds.sum(-2)
    dists = -0.5 * ((deltas * (-log_stds).exp())**2).sum(-2)
    log_probs = torch.log_softmax(logit_pi, -1) + log_norm + dists
    return torch.logsumexp(log_probs, dim=-1, keepdim=True)
