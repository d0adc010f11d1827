"""CPU oracle for the discrete-diffusion machinery.  TEST INFRASTRUCTURE ONLY.

Restates, in plain numpy / torch-CPU:
  cosine_beta_schedule_discrete        reference mb_vae_dti/training/diffusion/utils.py:74-86
  PredefinedNoiseScheduleDiscrete      diffusion/discrete_noise.py:17-44
  MarginalUniformTransition            diffusion/discrete_noise.py:47-99
  sample_discrete_features             diffusion/utils.py:124-162
  sample_discrete_feature_noise        diffusion/utils.py:284-317
  compute_posterior_distribution       diffusion/utils.py:258-280
  compute_batched_over0_posterior_...  diffusion/utils.py:319-349
  mask_distributions                   diffusion/utils.py:200-249
  FullDTIModel.apply_noise             training/modules/full.py:451-509
  FullDTIModel.sample_p_zs_given_zt    training/modules/full.py:1120-1171 (after the network call)
  FullDTIModel.kl_prior / compute_Lt   training/modules/full.py:569-677
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file.  Pinned by tests/golden/ fixtures captured from the reference.

Randomness contract: `p.multinomial(1)` on the reference side is reproduced as
argmax_k(p[k] / q[k]) with q ~ Exp(1) drawn by `torch.empty_like(p).exponential_(1)`
from the same generator state (that is how ATen implements multinomial without
replacement for one sample); every sampler here therefore takes the Exp(1)
tensors explicitly.
"""
import numpy as np
import torch
import torch.nn.functional as F


# --------------------------------------------------------------------------- schedule
def cosine_betas(T, s=0.008):
    steps = T + 2
    x = np.linspace(0, steps, steps)
    ac = np.cos(0.5 * np.pi * ((x / steps) + s) / (1 + s)) ** 2
    ac = ac / ac[0]
    return (1.0 - ac[1:] / ac[:-1]).squeeze()


class Schedule:
    """betas (float32, length T+1, unclamped) and alphas_bar built from clamped betas."""

    def __init__(self, T):
        self.T = T
        self.betas = torch.from_numpy(cosine_betas(T)).float()
        alphas = 1 - torch.clamp(self.betas, min=0, max=0.9999)
        self.alphas_bar = torch.exp(torch.cumsum(torch.log(alphas), dim=0))

    @staticmethod
    def _index(T, t_normalized=None, t_int=None):
        if t_int is None:
            t_int = torch.round(t_normalized * T)
        return t_int.long()

    def beta(self, t_normalized=None, t_int=None):
        return self.betas[self._index(self.T, t_normalized, t_int)]

    def alpha_bar(self, t_normalized=None, t_int=None):
        # negative indices wrap (t=0 in training gives s=-1 -> alphas_bar[T])
        return self.alphas_bar[self._index(self.T, t_normalized, t_int)]


def Qt(beta_t, marginals):
    """One-step transition (1-beta) I + beta 1 m^T for a batch of betas [bs,1] -> [bs,c,c]."""
    c = marginals.numel()
    b = beta_t.reshape(-1, 1, 1)
    return b * marginals.reshape(1, 1, c).expand(1, c, c) + (1 - b) * torch.eye(c).unsqueeze(0)


def Qt_bar(alpha_bar_t, marginals):
    """t-step transition abar I + (1-abar) 1 m^T -> [bs,c,c]."""
    c = marginals.numel()
    a = alpha_bar_t.reshape(-1, 1, 1)
    return a * torch.eye(c).unsqueeze(0) + (1 - a) * marginals.reshape(1, 1, c).expand(1, c, c)


# --------------------------------------------------------------------------- categorical sampling
def categorical_from_exp(prob, expo):
    """argmax_k prob[k]/expo[k] over the last axis (first index on ties)."""
    return torch.argmax(prob / expo, dim=-1)


def sample_graph(probX, probE, node_mask, expX, expE):
    """Class ids X_t [bs,n], E_t [bs,n,n] (int64), E symmetric with zero diagonal.
    Masked nodes / pairs and the diagonal are set to the uniform distribution before
    drawing, all n*n pairs are drawn, the strict upper triangle is kept and mirrored."""
    bs, n, cx = probX.shape
    ce = probE.shape[-1]
    probX = probX.clone()
    probE = probE.clone()
    probX[~node_mask] = 1.0 / cx
    pair_ok = node_mask.unsqueeze(1) & node_mask.unsqueeze(2)
    probE[~pair_ok] = 1.0 / ce
    probE[:, torch.arange(n), torch.arange(n)] = 1.0 / ce
    X_t = categorical_from_exp(probX, expX.reshape(bs, n, cx))
    E_t = categorical_from_exp(probE, expE.reshape(bs, n, n, ce))
    E_t = torch.triu(E_t, diagonal=1)
    E_t = E_t + E_t.transpose(1, 2)
    return X_t, E_t


def one_hot_masked(X_cls, E_cls, node_mask, cx, ce):
    xm = node_mask.unsqueeze(-1).float()
    X = F.one_hot(X_cls, cx).float() * xm
    E = F.one_hot(E_cls, ce).float() * (xm.unsqueeze(2) * xm.unsqueeze(1))
    return X, E


def apply_noise(sched, mx, me, X, E, node_mask, t_int, expX, expE):
    """Forward noising for given integer timesteps t_int [bs,1] (float) and Exp(1) draws."""
    T = sched.T
    s_int = t_int - 1
    t = t_int / T
    s = s_int / T
    beta_t = sched.beta(t_normalized=t)
    a_s = sched.alpha_bar(t_normalized=s)
    a_t = sched.alpha_bar(t_normalized=t)
    probX = X @ Qt_bar(a_t, mx)
    probE = E @ Qt_bar(a_t, me).unsqueeze(1)
    Xc, Ec = sample_graph(probX, probE, node_mask, expX, expE)
    X_t, E_t = one_hot_masked(Xc, Ec, node_mask, mx.numel(), me.numel())
    return X_t, E_t, {"t_int": t_int, "t": t, "beta_t": beta_t, "alpha_s_bar": a_s, "alpha_t_bar": a_t}, Xc, Ec


def sample_limit(mx, me, node_mask, expX, expE):
    """z_T ~ limit distribution; strict upper triangle kept and mirrored, then masked."""
    bs, n = node_mask.shape
    Xc = categorical_from_exp(mx.expand(bs, n, -1), expX.reshape(bs, n, -1))
    Ec = categorical_from_exp(me.expand(bs, n, n, -1), expE.reshape(bs, n, n, -1))
    X = F.one_hot(Xc, mx.numel()).float()
    E = F.one_hot(Ec, me.numel()).float()
    upper = torch.triu(torch.ones(n, n), diagonal=1)[None, :, :, None]
    E = E * upper
    E = E + E.transpose(1, 2)
    xm = node_mask.unsqueeze(-1).float()
    return X * xm, E * (xm.unsqueeze(2) * xm.unsqueeze(1))


# --------------------------------------------------------------------------- posteriors
def posterior(M, M_t, Qt_M, Qsb_M, Qtb_M):
    """q(z_s | z_t, x) up to the reference's conventions: (z_t Qt^T * x Qsb) / (x Qtb z_t^T)."""
    M = M.flatten(1, -2).float()
    M_t = M_t.flatten(1, -2).float()
    num = (M_t @ Qt_M.transpose(-2, -1)) * (M @ Qsb_M)
    den = ((M @ Qtb_M) * M_t).sum(-1)
    den = torch.where(den == 0, torch.full_like(den, 1e-6), den)
    return num / den.unsqueeze(-1)


def posterior_over_x0(Z_t, Qt_M, Qsb_M, Qtb_M):
    """[bs,N,c0,c_s]: for every candidate clean class x0, p(z_s, z_t | x0) / p(z_t | x0)."""
    Z = Z_t.flatten(1, -2).float()
    left = (Z @ Qt_M.transpose(-1, -2)).unsqueeze(2)           # bs,N,1,cs
    num = left * Qsb_M.unsqueeze(1)                            # bs,N,c0,cs
    den = (Qtb_M @ Z.transpose(-1, -2)).transpose(-1, -2).unsqueeze(-1)
    den = torch.where(den == 0, torch.full_like(den, 1e-6), den)
    return num / den


def reverse_step_probs(pred_X_logits, pred_E_logits, X_t, E_t, beta_t, a_s, a_t, mx, me):
    """prob_X [bs,n,cx], prob_E [bs,n,n,ce] of p(z_s | z_t) given network logits."""
    bs, n = X_t.shape[:2]
    out = []
    for logits, Z, m in ((pred_X_logits, X_t, mx), (pred_E_logits, E_t, me)):
        p0 = torch.softmax(logits, dim=-1).reshape(bs, -1, m.numel())
        post = posterior_over_x0(Z, Qt(beta_t, m), Qt_bar(a_s, m), Qt_bar(a_t, m))
        un = (p0.unsqueeze(-1) * post).sum(dim=2)
        zero = un.sum(-1) == 0
        un = torch.where(zero.unsqueeze(-1), torch.full_like(un, 1e-5), un)
        out.append(un / un.sum(-1, keepdim=True))
    return out[0], out[1].reshape(bs, n, n, -1)


def mask_distributions(true_X, true_E, pred_X, pred_E, node_mask):
    n = node_mask.shape[1]
    def fix(t, bad):
        t = t.clone()
        row = torch.zeros(t.shape[-1]); row[0] = 1.0
        t[bad] = row
        t = t + 1e-7
        return t / t.sum(-1, keepdim=True)
    bad_x = ~node_mask
    offdiag = ~torch.eye(n, dtype=torch.bool).unsqueeze(0)
    bad_e = ~(node_mask.unsqueeze(1) & node_mask.unsqueeze(2) & offdiag)
    return fix(true_X, bad_x), fix(true_E, bad_e), fix(pred_X, bad_x), fix(pred_E, bad_e)


def kl_prior(sched, mx, me, X, E, node_mask):
    bs, n = X.shape[:2]
    a_T = sched.alpha_bar(t_int=torch.full((bs, 1), float(sched.T)))
    probX = X @ Qt_bar(a_T, mx)
    probE = E @ Qt_bar(a_T, me).unsqueeze(1)
    lim_X = mx.expand(bs, n, -1)
    lim_E = me.expand(bs, n, n, -1)
    lX, lE, pX, pE = mask_distributions(lim_X, lim_E, probX, probE, node_mask)
    kx = F.kl_div(pX.log(), lX, reduction="none").reshape(bs, -1).sum(-1)
    ke = F.kl_div(pE.log(), lE, reduction="none").reshape(bs, -1).sum(-1)
    return kx + ke


def compute_Lt_terms(T, mx, me, X, E, X_t, E_t, pred_X_logits, pred_E_logits, node_mask, noise):
    """Returns the masked (true, pred) posterior distributions and T*(KL_X+KL_E) averaged
    over the batch as SumExceptBatchKL does for a single update (graph_metrics.py:33-44)."""
    bs, n = X.shape[:2]
    pX = torch.softmax(pred_X_logits, -1)
    pE = torch.softmax(pred_E_logits, -1)
    res = {}
    for name, m, clean, pred, Z in (("X", mx, X, pX, X_t), ("E", me, E, pE, E_t)):
        q_t, q_sb, q_tb = Qt(noise["beta_t"], m), Qt_bar(noise["alpha_s_bar"], m), Qt_bar(noise["alpha_t_bar"], m)
        res["true_" + name] = posterior(clean, Z, q_t, q_sb, q_tb)
        res["pred_" + name] = posterior(pred, Z, q_t, q_sb, q_tb)
    tE = res["true_E"].reshape(bs, n, n, -1)
    pE_ = res["pred_E"].reshape(bs, n, n, -1)
    tX, tE, pX_, pE_ = mask_distributions(res["true_X"], tE, res["pred_X"], pE_, node_mask)
    kl_x = F.kl_div(torch.log(torch.clamp(pX_, min=1e-8)), tX, reduction="sum") / bs
    kl_e = F.kl_div(torch.log(torch.clamp(pE_, min=1e-8)), tE, reduction="sum") / bs
    return tX, tE, pX_, pE_, T * (kl_x + kl_e)


# --------------------------------------------------------------------------- ELBO estimator (validation / test)
def reconstruction_probs(sched, mx, me, X, E, node_mask, expX, expE, denoise):
    """t = 0 term of the ELBO (reference mb_vae_dti/training/modules/full.py:680-761): draw G_0 ~ q(G_0 | G) with
    Q_0 = (1 - beta_0) I + beta_0 1 m^T, one-hot it WITHOUT masking, let `denoise(X0, E0, t0)` return logits, softmax,
    and set masked nodes / masked pairs / the diagonal to all-ones rows.  Returns (probX0, probE0, X0, E0)."""
    bs, n = node_mask.shape
    t0 = torch.zeros(bs, 1)
    beta0 = sched.beta(t_normalized=t0)
    probX = X @ Qt(beta0, mx)
    probE = E @ Qt(beta0, me).unsqueeze(1)
    Xc, Ec = sample_graph(probX, probE, node_mask, expX, expE)
    X0, E0 = F.one_hot(Xc, mx.numel()).float(), F.one_hot(Ec, me.numel()).float()
    lX, lE = denoise(X0, E0, t0)
    pX, pE = torch.softmax(lX, -1), torch.softmax(lE, -1)
    pair = node_mask.unsqueeze(1) & node_mask.unsqueeze(2) & ~torch.eye(n, dtype=torch.bool).unsqueeze(0)
    pX = torch.where(node_mask.unsqueeze(-1), pX, torch.ones_like(pX))
    pE = torch.where(pair.unsqueeze(-1), pE, torch.ones_like(pE))
    return pX, pE, X0, E0


def val_nll(sched, mx, me, node_hist_prob, X, E, X_t, E_t, pred_X_logits, pred_E_logits, node_mask, noise, probX0, probE0):
    """Batch NLL estimate and its parts (reference full.py:764-813 with the single-update metric semantics of
    metrics/graph_metrics.py:13-65): -log p(N) + KL prior + T * KL(posteriors) - E_q log p(G | G_0), where the two
    batch-level terms are scalars broadcast over the per-graph ones, then averaged."""
    bs = X.shape[0]
    N = node_mask.sum(1).long()
    log_pN = torch.log(node_hist_prob[N] + 1e-6)
    klp = kl_prior(sched, mx, me, X, E, node_mask)
    *_, Lt = compute_Lt_terms(sched.T, mx, me, X, E, X_t, E_t, pred_X_logits, pred_E_logits, node_mask, noise)
    term0 = ((X * torch.clamp(probX0, min=1e-8).log()).sum() + (E * torch.clamp(probE0, min=1e-8).log()).sum()) / bs
    nlls = -log_pN + klp + Lt - term0
    return nlls.mean(), {"kl_prior": klp, "Lt": Lt, "log_pN": log_pN, "loss_term_0": term0}
