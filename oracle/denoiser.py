"""CPU oracle for the graph-transformer denoiser.  TEST INFRASTRUCTURE ONLY.

A functional (state-dict driven) restatement in plain torch-CPU ops of
  GraphTransformer.forward      reference mb_vae_dti/training/models/graph_transformer.py:310-343
  XEyTransformerLayer.forward   :114-146
  NodeEdgeBlock.forward         :187-267   (masked_softmax :22-27)
  Xtoy / Etoy                   :34-67
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this file; the product path never does.  Parity is pinned by the fixtures under
tests/golden/ which tools/gen_golden.py captured from the reference itself.

All functions take `p`, a mapping name -> tensor holding the reference's
state_dict entries for the module (keys relative to the module, e.g.
"tf_layers.0.self_attn.q.weight"), so one set of weights can drive the
reference, the oracle and the HIP path.
"""
import math

import torch
import torch.nn.functional as F


def _lin(p, name, x):
    return F.linear(x, p[name + ".weight"], p.get(name + ".bias"))


def _sub(p, prefix):
    n = len(prefix)
    return {k[n:]: v for k, v in p.items() if k.startswith(prefix)}


# --------------------------------------------------------------------------- pooling
def pool_stats(t, dims):
    """mean | min | max | unbiased std over `dims`, concatenated on the feature axis.
    Padded zeros and the diagonal are NOT excluded (graph_transformer.py:42-45, 61-64)."""
    mean = t.mean(dim=dims)
    mn = t.amin(dim=dims)
    mx = t.amax(dim=dims)
    sd = t.std(dim=dims)
    return torch.cat((mean, mn, mx, sd), dim=-1)


def x_to_y(p, X):
    return _lin(p, "lin", pool_stats(X, (1,)))


def e_to_y(p, E):
    return _lin(p, "lin", pool_stats(E, (1, 2)))


# --------------------------------------------------------------------------- edge attention
def edge_attention(Q, K, V, E1, E2, ye_add, ye_mul, node_mask, n_head):
    """The n x n x dx core of NodeEdgeBlock (graph_transformer.py:206-249).

    Q,K,V: [bs,n,dx] (already masked); E1,E2: [bs,n,n,dx] (already masked);
    ye_add, ye_mul: [bs,dx].  Returns (newE_pre [bs,n,n,dx], wV [bs,n,dx]).
    Orientation as in the reference code (its inline shape comments are swapped):
    Q.unsqueeze(2) is [bs,n,1,..], K.unsqueeze(1) is [bs,1,n,..], so
    Y[b,i,j] = Q[b,i] * K[b,j]; the softmax runs over j (keys), independently
    for every one of the dx channels; wV[b,i] = sum_j attn[b,i,j] * V[b,j]."""
    bs, n, dx = Q.shape
    df = dx // n_head
    Y = Q.unsqueeze(2) * K.unsqueeze(1)
    Y = Y / math.sqrt(df)
    Y = Y * (E1 + 1.0) + E2
    newE_pre = ye_add[:, None, None, :] + (ye_mul[:, None, None, :] + 1.0) * Y
    if node_mask.sum() == 0:
        attn = Y
    else:
        key_ok = node_mask[:, None, :, None]
        attn = torch.softmax(Y.masked_fill(~key_ok.expand_as(Y), float("-inf")), dim=2)
    wV = (attn * V.unsqueeze(1)).sum(dim=2)
    return newE_pre, wV


def node_edge_block(p, X, E, y, node_mask, n_head):
    xm = node_mask.unsqueeze(-1).to(X.dtype)
    em = (xm.unsqueeze(2) * xm.unsqueeze(1))        # [bs,n,n,1]
    Q = _lin(p, "q", X) * xm
    K = _lin(p, "k", X) * xm
    V = _lin(p, "v", X) * xm
    E1 = _lin(p, "e_mul", E) * em
    E2 = _lin(p, "e_add", E) * em
    newE_pre, wV = edge_attention(Q, K, V, E1, E2, _lin(p, "y_e_add", y), _lin(p, "y_e_mul", y),
                                  node_mask, n_head)
    newE = _lin(p, "e_out", newE_pre) * em
    newX = _lin(p, "x_out", _lin(p, "y_x_add", y).unsqueeze(1)
                + (_lin(p, "y_x_mul", y).unsqueeze(1) + 1.0) * wV) * xm
    new_y = _lin(p, "y_y", y) + x_to_y(_sub(p, "x_y."), X) + e_to_y(_sub(p, "e_y."), E)
    new_y = _lin(p, "y_out.2", F.relu(_lin(p, "y_out.0", new_y)))
    return newX, newE, new_y


# --------------------------------------------------------------------------- layer
def _ln(p, name, x, eps=1e-5):
    return F.layer_norm(x, (x.shape[-1],), p[name + ".weight"], p[name + ".bias"], eps)


def xey_layer(p, X, E, y, node_mask, n_head):
    """Eval-mode (dropout = identity) XEyTransformerLayer."""
    newX, newE, new_y = node_edge_block(_sub(p, "self_attn."), X, E, y, node_mask, n_head)
    X = _ln(p, "normX1", X + newX)
    E = _ln(p, "normE1", E + newE)
    y = _ln(p, "norm_y1", y + new_y)
    X = _ln(p, "normX2", X + _lin(p, "linX2", F.relu(_lin(p, "linX1", X))))
    E = _ln(p, "normE2", E + _lin(p, "linE2", F.relu(_lin(p, "linE1", E))))
    y = _ln(p, "norm_y2", y + _lin(p, "lin_y2", F.relu(_lin(p, "lin_y1", y))))
    return X, E, y


# --------------------------------------------------------------------------- whole network
def graph_transformer(p, X, E, y, node_mask, n_layers, n_head, out_dims):
    """Eval-mode GraphTransformer.forward; returns (X, E, y) already masked."""
    bs, n = X.shape[:2]
    xm = node_mask.unsqueeze(-1).to(X.dtype)
    em = xm.unsqueeze(2) * xm.unsqueeze(1)
    X_skip, E_skip, y_skip = X[..., :out_dims["X"]], E[..., :out_dims["E"]], y[..., :out_dims["y"]]

    def mlp2(prefix, t, last_act):
        h = F.relu(_lin(p, prefix + ".0", t))
        h = _lin(p, prefix + ".2", h)
        return F.relu(h) if last_act else h

    hE = mlp2("mlp_in_E", E, True)
    hE = (hE + hE.transpose(1, 2)) / 2
    hX = mlp2("mlp_in_X", X, True) * xm
    hE = hE * em
    hy = mlp2("mlp_in_y", y, True)
    for l in range(n_layers):
        hX, hE, hy = xey_layer(_sub(p, f"tf_layers.{l}."), hX, hE, hy, node_mask, n_head)
    oX = mlp2("mlp_out_X", hX, False) + X_skip
    oE = mlp2("mlp_out_E", hE, False) + E_skip
    oy = mlp2("mlp_out_y", hy, False) + y_skip
    offdiag = (~torch.eye(n, dtype=torch.bool, device=E.device))[None, :, :, None].to(oE.dtype)
    oE = oE * offdiag
    oE = 0.5 * (oE + oE.transpose(1, 2))
    return oX * xm, oE * em, oy
