"""CPU oracle for the extra graph / molecular features.  TEST INFRASTRUCTURE ONLY.

Restates reference mb_vae_dti/training/diffusion/augmentation.py:
  ExtraFeatures.__call__            :16-35
  NodeCycleFeatures / KNodeCycles   :38-51, :177-250
  EigenFeatures, compute_laplacian  :54-105
  get_eigenvalues_features          :108-124
  get_eigenvectors_features         :127-154
  ExtraMolecularFeatures & co       :255-314
and FullDTIModel.get_extra_features (training/modules/full.py:512-538).
Pinned by tests/golden/ fixtures captured from the reference.  The symmetric
eigendecomposition is gauge dependent (signs, degenerate subspaces), so callers
compare eigenvalue features exactly-ish and eigenvector features up to sign.
"""
import torch


def cycle_features(A):
    """Per-node 3/4/5-cycle counts [bs,n,3] and per-graph 3/4/5/6-cycle counts [bs,4] from a
    float adjacency matrix A [bs,n,n] (no clipping / scaling)."""
    d = A.sum(-1)
    A2 = A @ A
    A3 = A2 @ A
    A4 = A3 @ A
    A5 = A4 @ A
    A6 = A5 @ A
    dg = lambda M: torch.diagonal(M, dim1=-2, dim2=-1)
    c3 = dg(A3)
    x3, y3 = c3 / 2, c3.sum(-1) / 6
    c4 = dg(A4) - d * (d - 1) - (A @ d.unsqueeze(-1)).sum(-1)
    x4, y4 = c4 / 2, c4.sum(-1) / 8
    tri = c3 / 2
    c5 = (dg(A5) - 2 * ((A2 * A) @ d.unsqueeze(-1)).squeeze(-1) - 4 * d * tri
          - 2 * (A @ tri.unsqueeze(-1)).squeeze(-1) + 10 * tri)
    x5, y5 = c5 / 2, c5.sum(-1) / 10
    d2 = dg(A2)
    c6 = (dg(A6).sum(-1) - 3 * dg(A3 ** 2).sum(-1) + 9 * (A * A2.pow(2)).sum((-2, -1))
          - 6 * (d2 * dg(A4)).sum(-1) + 6 * dg(A4).sum(-1) - 4 * dg(A3).sum(-1)
          + 4 * d2.pow(3).sum(-1) + 3 * A3.sum((-2, -1)) - 12 * d2.pow(2).sum(-1) + 4 * d2.sum(-1))
    y6 = c6 / 12
    return torch.stack((x3, x4, x5), -1), torch.stack((y3, y4, y5, y6), -1)


def laplacian_masked(E_t, node_mask):
    """Combinatorial Laplacian of the bonded-pairs adjacency; padded nodes get 2n on the diagonal."""
    m = node_mask.float()
    A = E_t[..., 1:].sum(-1).float() * m.unsqueeze(1) * m.unsqueeze(2)
    L = torch.diag_embed(A.sum(-1)) - A
    L = (L + L.transpose(1, 2)) / 2
    n = A.shape[-1]
    pad = 2 * n * torch.eye(n).unsqueeze(0) * (~node_mask).unsqueeze(1) * (~node_mask).unsqueeze(2)
    return L * m.unsqueeze(1) * m.unsqueeze(2) + pad


def eigen_features(E_t, node_mask, lcc_noise, k_vals=5, k_vecs=2):
    """(n_components [bs,1], eigvals [bs,5], not_lcc [bs,n,1], eigvecs [bs,n,2]).
    `lcc_noise` is the randn(bs,n) the reference adds on padded positions before torch.mode."""
    bs, n = node_mask.shape
    L = laplacian_masked(E_t, node_mask)
    w, V = torch.linalg.eigh(L)
    w = w / node_mask.sum(1, keepdim=True)
    V = V * node_mask.unsqueeze(2) * node_mask.unsqueeze(1)
    ncc = (w < 1e-5).sum(-1)
    ext = int(ncc.max()) + k_vals - n
    w_ext = torch.hstack((w, 2 * torch.ones(bs, ext))) if ext > 0 else w
    idx = torch.arange(k_vals).unsqueeze(0) + ncc.unsqueeze(1)
    vals = torch.gather(w_ext, 1, idx)

    first = torch.round(V[:, :, 0], decimals=3) * node_mask + lcc_noise * (~node_mask)
    mode = torch.mode(first, dim=1).values
    not_lcc = ((first != mode.unsqueeze(1)) & node_mask).unsqueeze(-1).float()

    ext = int(ncc.max()) + k_vecs - n
    V_ext = torch.cat((V, torch.zeros(bs, n, ext)), 2) if ext > 0 else V
    idx = (torch.arange(k_vecs)[None, None, :] + ncc[:, None, None]).expand(-1, n, -1)
    vecs = torch.gather(V_ext, 2, idx) * node_mask.unsqueeze(2)
    return ncc.unsqueeze(-1), vals, not_lcc, vecs


BOND_ORDERS = (0.0, 1.0, 2.0, 3.0, 1.5)


def molecular_features(X_t, E_t, valencies, atom_weights, max_weight, truncate_weight=False):
    """charge [bs,n], valency [bs,n], weight [bs,1].  NB the reference takes the *argmax* of
    one_hot*order, i.e. the class index where order>0 (aromatic counts 4, not 1.5) and 0 for
    no-bond -- an integer 'valency'.
    truncate_weight: FullDTIModel.apply_noise returns int64 one-hots (modules/full.py:500-507), and WeightFeature
    casts the summed atomic weight with `.type_as(X)` before dividing (diffusion/augmentation.py:311-314): on the
    train / validation path the weight is therefore truncated to an integer; the sampling and t = 0 paths pass
    float one-hots and keep the fraction."""
    orders = torch.tensor(BOND_ORDERS).reshape(1, 1, 1, -1)
    cur = (E_t * orders).argmax(-1).sum(-1)
    normal = (X_t * torch.tensor(valencies).reshape(1, 1, -1)).argmax(-1)
    charge = (normal - cur).float()
    valency = cur.float()
    weight = torch.tensor(atom_weights)[X_t.argmax(-1)].sum(-1, keepdim=True).float()
    if truncate_weight:
        weight = torch.trunc(weight)
    return charge, valency, weight / max_weight


def extra_features(X_t, E_t, t, node_mask, lcc_noise, max_n_nodes, valencies, atom_weights, max_weight, truncate_weight=False):
    """X_extra [bs,n,8], y_extra [bs,13] as fed to the denoiser."""
    A = E_t[..., 1:].sum(-1).float()
    xc, yc = cycle_features(A)
    xc = torch.clamp(xc * node_mask.unsqueeze(-1) / 10, max=1.0)
    yc = torch.clamp(yc / 10, max=1.0)
    ncc, vals, not_lcc, vecs = eigen_features(E_t, node_mask, lcc_noise)
    charge, valency, weight = molecular_features(X_t, E_t, valencies, atom_weights, max_weight, truncate_weight)
    n_frac = node_mask.sum(1, keepdim=True) / max_n_nodes
    X_extra = torch.cat((xc, not_lcc, vecs, charge.unsqueeze(-1), valency.unsqueeze(-1)), -1)
    y_extra = torch.cat((n_frac, yc, ncc.float(), vals, weight, t), 1)
    return X_extra, y_extra
