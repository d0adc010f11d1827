"""Discrete-diffusion machinery (host side): noise schedule, marginal transitions, samplers.

Mirrors the interface of mb_vae_dti/training/diffusion/discrete_noise.py:17-99 and of the sampling helpers in
mb_vae_dti/training/diffusion/utils.py:124-196,251-349; the per-node / per-pair arithmetic runs in the HIP
kernels (ops.apply_noise / ops.reverse_sample), the [bs]-sized schedule lookups stay in torch.
"""
import numpy as np
import torch

from . import ops
from .containers import PlaceHolder


def cosine_beta_schedule_discrete(timesteps, s=0.008):
    """Cosine schedule, float64, length timesteps + 1 (diffusion/utils.py:74-86)."""
    steps = timesteps + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    return (1 - ac[1:] / ac[:-1]).squeeze()


class PredefinedNoiseScheduleDiscrete(torch.nn.Module):
    """Lookup tables beta_t (unclamped, a state_dict buffer) and alpha_bar_t (from clamped betas)."""

    def __init__(self, timesteps=500):
        super().__init__()
        self.timesteps = timesteps
        self.register_buffer("betas", torch.from_numpy(cosine_beta_schedule_discrete(timesteps)).float())
        self.alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(self.alphas), dim=0))

    def _idx(self, t_normalized, t_int):
        assert int(t_normalized is None) + int(t_int is None) == 1
        if t_int is None:
            t_int = torch.round(t_normalized * self.timesteps)
        return t_int.long()

    def forward(self, t_normalized=None, t_int=None):
        return self.betas[self._idx(t_normalized, t_int)]

    def get_alpha_bar(self, t_normalized=None, t_int=None):
        idx = self._idx(t_normalized, t_int)
        if self.alphas_bar.device != idx.device:
            self.alphas_bar = self.alphas_bar.to(idx.device)
        return self.alphas_bar[idx]


class MarginalUniformTransition:
    """Q_t = (1-beta) I + beta 1 m^T and Qbar_t = abar I + (1-abar) 1 m^T for nodes and edges.
    (The reference also builds an unused [bs,dy,dy] matrix for y; that is omitted.)"""

    def __init__(self, x_marginals, e_marginals, y_classes):
        self.X_classes, self.E_classes, self.y_classes = len(x_marginals), len(e_marginals), y_classes
        self.x_marginals, self.e_marginals = x_marginals, e_marginals

    @staticmethod
    def _mix(w_eye, w_marg, marg, device):
        """w_eye * I + w_marg * 1 m^T per graph -> [bs, c, c] (weights given separately: 1 - (1 - b) != b in fp32)."""
        c = marg.numel()
        return (w_eye.to(device).reshape(-1, 1, 1) * torch.eye(c, device=device).unsqueeze(0)
                + w_marg.to(device).reshape(-1, 1, 1) * marg.to(device).reshape(1, 1, c).expand(1, c, c))

    def get_Qt(self, beta_t, device):
        return PlaceHolder(X=self._mix(1 - beta_t, beta_t, self.x_marginals, device),
                           E=self._mix(1 - beta_t, beta_t, self.e_marginals, device))

    def get_Qt_bar(self, alpha_bar_t, device):
        return PlaceHolder(X=self._mix(alpha_bar_t, 1 - alpha_bar_t, self.x_marginals, device),
                           E=self._mix(alpha_bar_t, 1 - alpha_bar_t, self.e_marginals, device))


class DistributionNodes:
    """Empirical distribution of graph sizes (diffusion/utils.py:166-196)."""

    def __init__(self, histogram):
        histogram = {int(k): int(v) for k, v in histogram.items()}
        prob = torch.zeros(max(histogram) + 1)
        for k, v in histogram.items():
            prob[k] = v
        self.prob = prob / prob.sum()
        self.m = torch.distributions.Categorical(prob)

    def sample_n(self, n_samples, device):
        return self.m.sample((n_samples,)).to(device)

    def log_prob(self, batch_n_nodes):
        assert batch_n_nodes.dim() == 1
        return torch.log(self.prob.to(batch_n_nodes.device)[batch_n_nodes] + 1e-6)


def exp_draws(rows, classes, device, generator=None):
    """Exp(1) draws exactly as ATen's multinomial(1) makes them: empty(rows, classes).exponential_(1)."""
    return torch.empty(rows, classes, device=device, dtype=torch.float32).exponential_(1, generator=generator)


def sample_discrete_features(probX, probE, node_mask, generator=None):
    """Class ids (X [bs,n], E [bs,n,n] symmetric, zero diagonal) from row-stochastic probX/probE: same draw
    order as the reference (all nodes, then all n*n pairs).  Runs on the GPU via argmax(p / Exp(1))."""
    bs, n, cx = probX.shape
    ce = probE.shape[-1]
    eX = exp_draws(bs * n, cx, probX.device, generator)
    eE = exp_draws(bs * n * n, ce, probX.device, generator)
    pX = torch.where(node_mask.unsqueeze(-1), probX, torch.full_like(probX, 1 / cx))
    X = torch.argmax(pX / eX.reshape(bs, n, cx), -1)
    pair = node_mask.unsqueeze(1) & node_mask.unsqueeze(2)
    pE = torch.where(pair.unsqueeze(-1), probE, torch.full_like(probE, 1 / ce))
    E = torch.triu(torch.argmax(pE / eE.reshape(bs, n, n, ce), -1), diagonal=1)
    return PlaceHolder(X=X, E=E + E.transpose(1, 2), y=torch.zeros(bs, 0, device=probX.device).long())


def sample_discrete_feature_noise(limit_dist, node_mask, generator=None, exp=None):
    """z_T ~ limit distribution as masked one-hot floats (diffusion/utils.py:284-317).  exp = (expX [bs*n,cx],
    expE [bs*n*n,ce]): Exp(1) draws to use instead of the generator's."""
    bs, n = node_mask.shape
    dev = node_mask.device
    mx, me = limit_dist.X.to(dev).float(), limit_dist.E.to(dev).float()
    if exp is not None:
        eX, eE = exp[0].to(dev), exp[1].to(dev)
    else:
        eX = exp_draws(bs * n, mx.numel(), dev, generator)
        eE = exp_draws(bs * n * n, me.numel(), dev, generator)
    # forward noising with abar = 0 from any clean graph IS the limit distribution: reuse the kernel
    X0 = torch.zeros(bs, n, mx.numel(), device=dev); X0[..., 0] = 1
    E0 = torch.zeros(bs, n, n, me.numel(), device=dev); E0[..., 0] = 1
    zero = torch.zeros(bs, device=dev)
    _, _, Xt, Et = ops.apply_noise(X0, E0, torch.ones_like(node_mask), zero, mx, me, eX, eE)
    # the limit sample keeps an all-zero diagonal (mirror of the strict upper triangle)
    Et = Et * (~torch.eye(n, dtype=torch.bool, device=dev))[None, :, :, None]
    return PlaceHolder(X=Xt, E=Et, y=torch.zeros(bs, 0, device=dev)).mask(node_mask)
