"""Encoders, aggregators and fusion over precomputed feature vectors, on the HIP linear / LayerNorm kernels.

Same constructors, parameter names (state_dict keys) and return values as
mb_vae_dti/training/models/blocks.py (ResidualEncoder :18-66, TransformerEncoder :69-129, ConcatAggregator :136-168,
AttentiveAggregator :171-207, CrossAttentionFusion :214-332).  The module containers (nn.Sequential of
LayerNorm / Linear / activation / Dropout) are kept only to own the parameters under the reference's key names;
forward() walks them and issues the fused kernels.
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


def _ln(ln, x):
    beta = ln.bias if ln.bias is not None else torch.zeros_like(ln.weight)
    return ops.ln_residual(x, None, ln.weight, beta, ln.eps)


def _act_linear(lin, x, activation):
    """activation(lin(x)); ReLU is fused into the GEMM epilogue, other activations run on the (small) output."""
    if isinstance(activation, nn.ReLU):
        return ops.linear(x, lin.weight, lin.bias, act=1)
    return activation(ops.linear(x, lin.weight, lin.bias))


def _lin(lin, x):
    return ops.linear(x, lin.weight, lin.bias)


def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def _mlp_block(seq, x, activation, training):
    """LayerNorm -> Linear -> act -> Dropout -> Linear   (indices 0,1,2,3,4 of the Sequential)."""
    h = _act_linear(seq[1], _ln(seq[0], x), activation)
    return _lin(seq[4], _drop(h, seq[3].p, training))


def _gated(seq, x, u, gate_in, activation, training):
    """x + u * gate(gate_in), gate = LayerNorm -> Linear -> act -> Linear -> Sigmoid -> Dropout (indices 0..5 of the
    Sequential); u None: x + x * gate.  Sigmoid, dropout, product and sum run as one kernel (ops.gate_update)."""
    z = _lin(seq[3], _act_linear(seq[1], _ln(seq[0], gate_in), activation))
    return ops.gate_update(x, u, z, seq[5].p, training)


class ResidualEncoder(nn.Module):
    """Pre-norm residual MLP encoder."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_layers, factor=4, dropout=0.1, bias=True, activation=nn.ReLU()):
        super().__init__()
        self.n_layers, self.activation = n_layers, activation
        self.input_proj = nn.Linear(input_dim, hidden_dim, bias=bias)
        self.layers = nn.ModuleList([
            nn.Sequential(nn.LayerNorm(hidden_dim, bias=bias), nn.Linear(hidden_dim, hidden_dim * factor, bias=bias),
                          self.activation, nn.Dropout(dropout), nn.Linear(hidden_dim * factor, hidden_dim, bias=bias))
            for _ in range(n_layers)])
        self.output_proj = nn.Sequential(nn.LayerNorm(hidden_dim, bias=bias), nn.Linear(hidden_dim, output_dim, bias=bias))

    def forward(self, x):
        x = _lin(self.input_proj, x)
        for blk in self.layers:
            x = x + _mlp_block(blk, x, self.activation, self.training)
        return _lin(self.output_proj[1], _ln(self.output_proj[0], x))


class TransformerEncoder(nn.Module):
    """Alternating sigmoid-gate ("attention") and MLP residual blocks on a single feature vector."""

    def __init__(self, input_dim, hidden_dim, output_dim, n_layers, factor=4, dropout=0.1, bias=True, activation=nn.ReLU()):
        super().__init__()
        self.n_layers, self.activation = n_layers, activation
        self.input_proj = nn.Linear(input_dim, hidden_dim, bias=bias)
        self.att_layers, self.mlp_layers = nn.ModuleList(), nn.ModuleList()
        for _ in range(n_layers):
            self.att_layers.append(nn.Sequential(
                nn.LayerNorm(hidden_dim, bias=bias), nn.Linear(hidden_dim, hidden_dim // 4, bias=bias), self.activation,
                nn.Linear(hidden_dim // 4, hidden_dim, bias=bias), nn.Sigmoid(), nn.Dropout(dropout)))
            self.mlp_layers.append(nn.Sequential(
                nn.LayerNorm(hidden_dim, bias=bias), nn.Linear(hidden_dim, hidden_dim * factor, bias=bias), self.activation,
                nn.Dropout(dropout), nn.Linear(hidden_dim * factor, hidden_dim, bias=bias)))
        self.output_proj = nn.Sequential(nn.LayerNorm(hidden_dim, bias=bias), nn.Linear(hidden_dim, output_dim, bias=bias))

    def forward(self, x):
        x = _lin(self.input_proj, x)
        for att, mlp in zip(self.att_layers, self.mlp_layers):
            x = _gated(att, x, None, x, self.activation, self.training)
            x = x + _mlp_block(mlp, x, self.activation, self.training)
        return _lin(self.output_proj[1], _ln(self.output_proj[0], x))


class ConcatAggregator(nn.Module):
    def __init__(self, input_dim, output_dim, n_features, dropout=0.1, activation=nn.ReLU()):
        super().__init__()
        self.activation = activation
        total = input_dim * n_features
        self.projection = nn.Sequential(nn.LayerNorm(total), nn.Linear(total, output_dim), self.activation, nn.Dropout(dropout))

    def forward(self, features):
        h = _act_linear(self.projection[1], _ln(self.projection[0], torch.cat(features, dim=-1)), self.activation)
        return _drop(h, self.projection[3].p, self.training)


class AttentiveAggregator(nn.Module):
    """Softmax-weighted sum over the views (weights from a 1x1 conv), then LayerNorm-Linear-act-Dropout."""

    def __init__(self, input_dim, output_dim, n_features, dropout=0.1, activation=nn.ReLU()):
        super().__init__()
        self.activation = activation
        self.to_attn_logits = nn.Conv1d(input_dim, 1, 1)
        self.projection = nn.Sequential(nn.LayerNorm(input_dim), nn.Linear(input_dim, output_dim), self.activation, nn.Dropout(dropout))

    def forward(self, features):
        stacked = torch.stack(features, dim=-1)                                        # [B, D, V]
        w = self.to_attn_logits.weight.reshape(1, -1, 1)
        logits = (stacked * w).sum(1, keepdim=True) + self.to_attn_logits.bias.reshape(1, 1, 1)
        attn = F.softmax(logits, dim=-1)                                               # [B, 1, V]
        agg = (stacked * attn).sum(dim=-1)
        h = _act_linear(self.projection[1], _ln(self.projection[0], agg), self.activation)
        return _drop(h, self.projection[3].p, self.training), attn


class CrossAttentionFusion(nn.Module):
    """n_layers x (bidirectional gated cross update, self gate, MLP) on the drug / target pair, then attentive
    aggregation; returns (fused, attention weights)."""

    def __init__(self, input_dim, output_dim, n_layers, factor=4, dropout=0.1, bias=True, activation=nn.ReLU()):
        super().__init__()
        self.n_layers, self.activation = n_layers, activation
        names = ("cross_gates_a_to_b", "cross_gates_b_to_a", "align_proj_b_to_a", "align_proj_a_to_b", "self_gates_a",
                 "self_gates_b", "mlp_a", "mlp_b")
        for nm in names:
            setattr(self, nm, nn.ModuleList())

        def gate(width):
            return nn.Sequential(nn.LayerNorm(width, bias=bias), nn.Linear(width, input_dim // factor, bias=bias), self.activation,
                                 nn.Linear(input_dim // factor, input_dim, bias=bias), nn.Sigmoid(), nn.Dropout(dropout))

        def mlp():
            return nn.Sequential(nn.LayerNorm(input_dim, bias=bias), nn.Linear(input_dim, input_dim // factor, bias=bias),
                                 self.activation, nn.Dropout(dropout), nn.Linear(input_dim // factor, input_dim, bias=bias))

        for _ in range(n_layers):
            self.align_proj_b_to_a.append(nn.Linear(input_dim, input_dim, bias=False))
            self.align_proj_a_to_b.append(nn.Linear(input_dim, input_dim, bias=False))
            self.cross_gates_a_to_b.append(gate(input_dim * 2))
            self.cross_gates_b_to_a.append(gate(input_dim * 2))
            self.self_gates_a.append(gate(input_dim))
            self.self_gates_b.append(gate(input_dim))
            self.mlp_a.append(mlp())
            self.mlp_b.append(mlp())
        self.att_agg = AttentiveAggregator(input_dim=input_dim, output_dim=output_dim, n_features=2, dropout=dropout,
                                           activation=activation)

    def forward(self, a, b):
        act, tr = self.activation, self.training
        for i in range(self.n_layers):
            a, b = (_gated(self.cross_gates_b_to_a[i], a, _lin(self.align_proj_b_to_a[i], b), torch.cat([a, b], dim=-1), act, tr),
                    _gated(self.cross_gates_a_to_b[i], b, _lin(self.align_proj_a_to_b[i], a), torch.cat([b, a], dim=-1), act, tr))
            a = _gated(self.self_gates_a[i], a, None, a, act, tr)
            b = _gated(self.self_gates_b[i], b, None, b, act, tr)
            a = a + _mlp_block(self.mlp_a[i], a, act, tr)
            b = b + _mlp_block(self.mlp_b[i], b, act, tr)
        return self.att_agg([a, b])
