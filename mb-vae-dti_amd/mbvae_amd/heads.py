"""Prediction / loss heads on the HIP kernels.

Same constructors, parameter names and return values as mb_vae_dti/training/models/heads.py
(DTIHead :12-57, InfoNCEHead :60-189, KLVariationalHead :192-241, ReconstructionHead :244-299).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops
from .blocks import _act_linear, _drop, _ln


class DTIHead(nn.Module):
    """LayerNorm-Linear-act-Dropout trunk and four scalar heads (binary Y, pKd, pKi, KIBA)."""

    def __init__(self, input_dim, proj_dim, dropout, bias, activation):
        super().__init__()
        self.activation = activation
        self.shared_layers = nn.Sequential(nn.LayerNorm(input_dim), nn.Linear(input_dim, proj_dim, bias=bias), self.activation,
                                           nn.Dropout(dropout))
        self.binary_head = nn.Linear(proj_dim, 1)
        self.pKd_head = nn.Linear(proj_dim, 1)
        self.pKi_head = nn.Linear(proj_dim, 1)
        self.KIBA_head = nn.Linear(proj_dim, 1)

    def forward(self, features):
        h = _act_linear(self.shared_layers[1], _ln(self.shared_layers[0], features), self.activation)
        h = _drop(h, self.shared_layers[3].p, self.training)
        # the four 1-wide heads run as ONE [proj_dim -> 4] product
        w = torch.cat([self.binary_head.weight, self.pKd_head.weight, self.pKi_head.weight, self.KIBA_head.weight], 0)
        b = torch.cat([self.binary_head.bias, self.pKd_head.bias, self.pKi_head.bias, self.KIBA_head.bias], 0)
        out = ops.linear(h, w, b)
        return {"Y": out[:, 0], "Y_pKd": out[:, 1], "Y_pKi": out[:, 2], "Y_KIBA": out[:, 3]}


class InfoNCEHead(nn.Module):
    """InfoNCE with the Tanimoto-nearest neighbour as positive and (1 - Tanimoto)-weighted negatives."""

    def __init__(self, input_dim, proj_dim, dropout, bias, activation):
        super().__init__()
        self.activation = activation
        self.projection = nn.Sequential(nn.LayerNorm(input_dim), nn.Linear(input_dim, proj_dim, bias=bias), self.activation,
                                        nn.Dropout(dropout))

    @staticmethod
    def tanimoto_similarity(fp):
        inter = fp @ fp.T
        s = fp.sum(dim=1)
        return inter / torch.clamp(s.unsqueeze(1) + s.unsqueeze(0) - inter, min=1e-8)

    def info_nce_loss(self, features, similarity_matrix, temperature):
        B = features.shape[0]
        z = F.normalize(features, dim=1)
        eye = torch.eye(B, device=z.device, dtype=torch.bool)
        logits = (z @ z.T / temperature).masked_fill(eye, float("-inf"))
        sim = similarity_matrix.masked_fill(eye, 0.0)
        positive = sim.argmax(dim=1)
        untouched = eye.clone()
        untouched[torch.arange(B, device=z.device), positive] = True
        weighted = torch.where(untouched, logits, logits * (1.0 - sim).to(logits.dtype))
        return F.cross_entropy(weighted, positive)

    def forward(self, x, fingerprints, temperature):
        z = _act_linear(self.projection[1], _ln(self.projection[0], x), self.activation)
        z = _drop(z, self.projection[3].p, self.training)
        return self.info_nce_loss(z, self.tanimoto_similarity(fingerprints), temperature)


class KLVariationalHead(nn.Module):
    """mu / logvar projections, reparameterised sample and KL to N(0, I) averaged over batch and latent dims."""

    def __init__(self, input_dim, proj_dim):
        super().__init__()
        self.fc_mu = nn.Linear(input_dim, proj_dim, bias=True)
        self.fc_logvar = nn.Linear(input_dim, proj_dim, bias=True)
        self._last_kl = None

    def reparameterize(self, mu, logvar, eps=None):
        eps = torch.randn_like(mu) if eps is None else eps
        z, kl = ops.kl_reparam(mu, logvar, eps)
        self._last_kl = (mu, logvar, kl)
        return z

    def kl_divergence(self, mu, logvar):
        """KL for (mu, logvar); reuses the value the fused kernel produced in forward() for the same tensors."""
        if self._last_kl is not None and self._last_kl[0] is mu and self._last_kl[1] is logvar:
            return self._last_kl[2]
        _, kl = ops.kl_reparam(mu, logvar, torch.zeros_like(mu))
        return kl

    def forward(self, x, eps=None):
        mu = ops.linear(x, self.fc_mu.weight, self.fc_mu.bias)
        logvar = ops.linear(x, self.fc_logvar.weight, self.fc_logvar.bias)
        return self.reparameterize(mu, logvar, eps), mu, logvar


class ReconstructionHead(nn.Module):
    """lambda_X * CE_X + lambda_E * CE_E over rows whose one-hot target is non-zero (padded pairs are labelled
    'no bond' by the dense-batch convention and therefore count)."""

    def __init__(self, diff_weights):
        super().__init__()
        assert len(diff_weights) == 2, "Reconstruction head expects 2 weights for X and E"
        self.diff_weights = diff_weights
        self.last_stats = None

    def forward(self, masked_pred_X, masked_pred_E, true_X, true_E):
        lX, sX = ops.masked_ce(masked_pred_X, true_X.float())
        lE, sE = ops.masked_ce(masked_pred_E, true_E.float())
        self.last_stats = (sX, sE)       # [sum CE, #rows, per-class BCE sums]: feeds the train-time per-class trackers
        return lX * self.diff_weights[0] + lE * self.diff_weights[1]
