"""Extra structural / molecular features appended to the denoiser input (X: +8, y: +13).

Mirrors the callables of mb_vae_dti/training/diffusion/augmentation.py (ExtraFeatures :16-35,
ExtraMolecularFeatures :255-276) without their host-synchronising asserts and Python max() over tensors.
"""
import torch

from . import ops


def _diag(M):
    return torch.diagonal(M, dim1=-2, dim2=-1)


def cycle_counts(A):
    """Per-node 3/4/5-cycle and per-graph 3/4/5/6-cycle counts from adjacency A [bs,n,n] (augmentation.py:177-250).
    One HIP kernel for n <= 64; larger graphs take the same formulas as batched device matmuls."""
    if A.is_cuda and A.shape[-1] <= 64 and A.dtype == torch.float32:
        return ops.cycle_counts(A)
    d = A.sum(-1)
    A2 = A @ A
    A3 = A2 @ A
    A4 = A3 @ A
    A5 = A4 @ A
    A6 = A5 @ A
    c3 = _diag(A3)
    c4 = _diag(A4) - d * (d - 1) - (A @ d.unsqueeze(-1)).sum(-1)
    tri = c3 / 2
    c5 = (_diag(A5) - 2 * ((A2 * A) @ d.unsqueeze(-1)).squeeze(-1) - 4 * d * tri
          - 2 * (A @ tri.unsqueeze(-1)).squeeze(-1) + 10 * tri)
    d2 = _diag(A2)
    c6 = (_diag(A6).sum(-1) - 3 * _diag(A3 ** 2).sum(-1) + 9 * (A * A2.pow(2)).sum((-2, -1))
          - 6 * (d2 * _diag(A4)).sum(-1) + 6 * _diag(A4).sum(-1) - 4 * c3.sum(-1)
          + 4 * d2.pow(3).sum(-1) + 3 * A3.sum((-2, -1)) - 12 * d2.pow(2).sum(-1) + 4 * d2.sum(-1))
    x = torch.stack((c3 / 2, c4 / 2, c5 / 2), -1)
    y = torch.stack((c3.sum(-1) / 6, c4.sum(-1) / 8, c5.sum(-1) / 10, c6 / 12), -1)
    return x, y


class ExtraFeatures:
    def __init__(self, max_n_nodes=64):
        self.max_n_nodes = max_n_nodes

    def __call__(self, E_t, node_mask, lcc_noise=None, eigvec=None):
        """eigvec [bs,n,2] (optional) replaces the two eigenvector features: eigenvectors are only defined up to sign
        (and up to a rotation inside a degenerate eigenspace), so a replay of another implementation's step must use
        that implementation's choice."""
        bs, n = node_mask.shape
        m = node_mask.float()
        A_raw = E_t[..., 1:].sum(-1).float()
        xc, yc = cycle_counts(A_raw)
        xc = torch.clamp(xc * m.unsqueeze(-1) / 10, max=1.0)
        yc = torch.clamp(yc / 10, max=1.0)

        A = A_raw * m.unsqueeze(1) * m.unsqueeze(2)
        L = torch.diag_embed(A.sum(-1)) - A
        L = (L + L.transpose(1, 2)) / 2
        pad = 2 * n * torch.eye(n, device=A.device).unsqueeze(0) * (1 - m).unsqueeze(1) * (1 - m).unsqueeze(2)
        L = L * m.unsqueeze(1) * m.unsqueeze(2) + pad
        w, V = ops.sym_eig(L)
        w = w / m.sum(1, keepdim=True)
        V = V * m.unsqueeze(2) * m.unsqueeze(1)
        ncc = (w < 1e-5).sum(-1)
        # always extend by k columns: gather indices ncc + [0,k) never exceed n + k - 1
        k_val, k_vec = 5, 2
        w_ext = torch.cat((w, 2 * torch.ones(bs, k_val, device=w.device)), 1)
        vals = torch.gather(w_ext, 1, torch.arange(k_val, device=w.device).unsqueeze(0) + ncc.unsqueeze(1))
        noise = torch.randn(bs, n, device=m.device) if lcc_noise is None else lcc_noise
        first = torch.round(V[:, :, 0], decimals=3) * m + noise * (1 - m)
        mode = torch.mode(first, dim=1).values
        not_lcc = ((first != mode.unsqueeze(1)) & node_mask).unsqueeze(-1).float()
        V_ext = torch.cat((V, torch.zeros(bs, n, k_vec, device=V.device)), 2)
        idx = (torch.arange(k_vec, device=V.device)[None, None, :] + ncc[:, None, None]).expand(-1, n, -1)
        vecs = torch.gather(V_ext, 2, idx) * m.unsqueeze(2)
        self.last_eigvec = vecs
        if eigvec is not None:
            vecs = eigvec.to(vecs)

        X_extra = torch.cat((xc, not_lcc, vecs), -1)
        y_extra = torch.cat((m.sum(1, keepdim=True) / self.max_n_nodes, yc, ncc.unsqueeze(-1).float(), vals), 1)
        return X_extra, y_extra


class ExtraMolecularFeatures:
    def __init__(self, valencies, max_weight, atom_weights):
        self.valencies, self.max_weight, self.atom_weights = list(valencies), max_weight, list(atom_weights)
        self._const = {}

    def _constants(self, dev):
        """Bond orders / valencies / atomic weights as device tensors, built once per device (a host->device copy
        per call would synchronise every step and cannot be captured in a HIP graph)."""
        key = str(dev)
        if key not in self._const:
            self._const[key] = (torch.tensor([0, 1, 2, 3, 1.5], device=dev).reshape(1, 1, 1, -1),
                                torch.tensor(self.valencies, device=dev).reshape(1, 1, -1),
                                torch.tensor(self.atom_weights, device=dev))
        return self._const[key]

    def __call__(self, X_t, E_t, truncate_weight=False):
        """truncate_weight: the reference's training path hands int64 one-hot tensors to this feature, which makes
        its `.type_as(X_t)` truncate the summed atomic weight to an integer before the division by max_weight;
        the sampling path passes floats and keeps the fraction.  Callers pick the behaviour of the path they mirror."""
        orders, valencies, weights = self._constants(X_t.device)
        cur = (E_t * orders).argmax(-1).sum(-1)                               # integer "valency" as in the reference
        normal = (X_t * valencies).argmax(-1)
        charge = (normal - cur).type_as(X_t)
        weight = weights[X_t.argmax(-1)].sum(-1, keepdim=True)
        weight = (torch.trunc(weight) if truncate_weight else weight).type_as(X_t) / self.max_weight
        return torch.stack((charge, cur.type_as(X_t)), -1), weight
