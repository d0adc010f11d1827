"""CPU oracle for encoders, aggregators, fusion and loss heads.  TEST INFRASTRUCTURE ONLY.

State-dict driven, eval-mode (dropout = identity) restatements of
  ResidualEncoder / TransformerEncoder          reference mb_vae_dti/training/models/blocks.py:18-129
  ConcatAggregator / AttentiveAggregator        blocks.py:136-207
  CrossAttentionFusion                          blocks.py:214-332
  DTIHead / InfoNCEHead / KLVariationalHead     mb_vae_dti/training/models/heads.py:12-241
  ReconstructionHead                            heads.py:244-299
  CEPerClass (per-class BCE trackers)           mb_vae_dti/training/metrics/molecular_metrics.py:68-97
  DTI accuracy loss                             mb_vae_dti/training/modules/full.py:913-950
Pinned by tests/golden/ fixtures captured from the reference.
"""
import torch
import torch.nn.functional as F


def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def _ln(p, name, x):
    return F.layer_norm(x, (x.shape[-1],), p.get(name + ".weight"), p.get(name + ".bias"), 1e-5)


def _n_blocks(p, stem):
    return len({k.split(".")[1] for k in p if k.startswith(stem + ".")})


# --------------------------------------------------------------------------- encoders
def residual_encoder(p, x, act=F.relu):
    x = _lin(p, "input_proj", x)
    for i in range(_n_blocks(p, "layers")):
        h = _ln(p, f"layers.{i}.0", x)
        x = x + _lin(p, f"layers.{i}.4", act(_lin(p, f"layers.{i}.1", h)))
    return _lin(p, "output_proj.1", _ln(p, "output_proj.0", x))


def transformer_encoder(p, x, act=F.relu):
    x = _lin(p, "input_proj", x)
    for i in range(_n_blocks(p, "att_layers")):
        g = _lin(p, f"att_layers.{i}.3", act(_lin(p, f"att_layers.{i}.1", _ln(p, f"att_layers.{i}.0", x))))
        x = x + x * torch.sigmoid(g)
        h = _ln(p, f"mlp_layers.{i}.0", x)
        x = x + _lin(p, f"mlp_layers.{i}.4", act(_lin(p, f"mlp_layers.{i}.1", h)))
    return _lin(p, "output_proj.1", _ln(p, "output_proj.0", x))


# --------------------------------------------------------------------------- aggregators
def concat_aggregator(p, feats, act=F.relu):
    x = torch.cat(feats, -1)
    return act(_lin(p, "projection.1", _ln(p, "projection.0", x)))


def attentive_aggregator(p, feats, act=F.relu):
    st = torch.stack(feats, -1)                                    # [B, D, V]
    w = p["to_attn_logits.weight"].reshape(1, -1, 1)               # conv1d(D,1,1)
    logits = (st * w).sum(1, keepdim=True) + p["to_attn_logits.bias"].reshape(1, 1, 1)
    a = torch.softmax(logits, -1)
    agg = (st * a).sum(-1)
    return act(_lin(p, "projection.1", _ln(p, "projection.0", agg))), a


def _gate(p, name, x, act):
    return torch.sigmoid(_lin(p, name + ".3", act(_lin(p, name + ".1", _ln(p, name + ".0", x)))))


def _mlp(p, name, x, act):
    return _lin(p, name + ".4", act(_lin(p, name + ".1", _ln(p, name + ".0", x))))


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


def cross_attention_fusion(p, a, b, act=F.relu):
    for i in range(_n_blocks(p, "mlp_a")):
        ga = _gate(p, f"cross_gates_b_to_a.{i}", torch.cat([a, b], -1), act)
        gb = _gate(p, f"cross_gates_a_to_b.{i}", torch.cat([b, a], -1), act)
        a, b = (a + ga * F.linear(b, p[f"align_proj_b_to_a.{i}.weight"]),
                b + gb * F.linear(a, p[f"align_proj_a_to_b.{i}.weight"]))
        a = a + a * _gate(p, f"self_gates_a.{i}", a, act)
        b = b + b * _gate(p, f"self_gates_b.{i}", b, act)
        a = a + _mlp(p, f"mlp_a.{i}", a, act)
        b = b + _mlp(p, f"mlp_b.{i}", b, act)
    return attentive_aggregator(_sub(p, "att_agg."), [a, b], act)


# --------------------------------------------------------------------------- heads
def dti_head(p, x, act=F.relu):
    h = act(_lin(p, "shared_layers.1", _ln(p, "shared_layers.0", x)))
    return {"Y": _lin(p, "binary_head", h).squeeze(-1), "Y_pKd": _lin(p, "pKd_head", h).squeeze(-1),
            "Y_pKi": _lin(p, "pKi_head", h).squeeze(-1), "Y_KIBA": _lin(p, "KIBA_head", h).squeeze(-1)}


def kl_head(p, x, eps):
    mu, logvar = _lin(p, "fc_mu", x), _lin(p, "fc_logvar", x)
    return mu + eps * torch.exp(0.5 * logvar), mu, logvar


def kl_divergence(mu, logvar):
    return -0.5 * torch.sum(1 + logvar - mu.pow(2) - logvar.exp()) / mu.numel()


def tanimoto(fp):
    inter = fp @ fp.T
    s = fp.sum(1)
    return inter / torch.clamp(s[:, None] + s[None, :] - inter, min=1e-8)


def info_nce(p, x, fingerprints, temperature, act=F.relu):
    z = act(_lin(p, "projection.1", _ln(p, "projection.0", x)))
    B = z.shape[0]
    z = F.normalize(z, dim=1)
    logits = (z @ z.T) / temperature
    eye = torch.eye(B, dtype=torch.bool)
    logits = logits.masked_fill(eye, float("-inf"))
    sim = tanimoto(fingerprints).masked_fill(eye, 0.0)
    pos = sim.argmax(1)
    keep = eye.clone()
    keep[torch.arange(B), pos] = True
    weighted = torch.where(keep, logits, logits * (1.0 - sim))
    return F.cross_entropy(weighted, pos)


def reconstruction_loss(pred_X, pred_E, true_X, true_E, lambdas):
    """Rows whose one-hot target is all-zero are dropped; CE summed / #kept rows, per X and E."""
    out = []
    for pred, true in ((pred_X, true_X), (pred_E, true_E)):
        true = true.reshape(-1, true.shape[-1])
        pred = pred.reshape(-1, pred.shape[-1])
        keep = (true != 0).any(-1)
        if keep.any():
            out.append(F.cross_entropy(pred[keep], true[keep].argmax(-1), reduction="sum") / keep.sum())
        else:
            out.append(torch.zeros(()))
    return out[0] * lambdas[0] + out[1] * lambdas[1]


def per_class_bce(pred, true):
    """sum over kept rows of BCEWithLogits(pred[:,c], true[:,c]) for every class c, and #kept rows."""
    true = true.reshape(-1, true.shape[-1])
    pred = pred.reshape(-1, pred.shape[-1])
    keep = (true != 0).any(-1)
    tot = F.binary_cross_entropy_with_logits(pred[keep], true[keep], reduction="none").sum(0)
    return tot, keep.sum()


def dti_accuracy_loss(scores, targets, masks, dti_weights):
    comps = [F.binary_cross_entropy_with_logits(scores["Y"], targets["Y"])]
    for name in ("Y_pKd", "Y_pKi", "Y_KIBA"):
        m = masks[name]
        comps.append(F.mse_loss(scores[name][m], targets[name][m]) if m.any() else torch.zeros(()))
    return sum(w * c for w, c in zip(dti_weights, comps)), comps
