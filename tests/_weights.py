"""Deterministic, name-seeded synthetic weights shared by the fixture generator and the tests
(for modules whose state dict is too large to commit as a fixture)."""
import zlib

import torch


def synthetic_state_dict(shapes, scale=None):
    """shapes: {state_dict key: shape tuple}.  Linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.05^2),
    LayerNorm weights ~ 1 + N(0, 0.1^2); every tensor drawn from its own generator seeded by crc32(key)."""
    out = {}
    for name, shape in shapes.items():
        g = torch.Generator().manual_seed(zlib.crc32(name.encode()))
        t = torch.randn(*shape, generator=g)
        if "norm" in name and name.endswith("weight"):
            t = 1.0 + 0.1 * t
        elif name.endswith("weight") and len(shape) == 2:
            t = t / (shape[1] ** 0.5)
        else:
            t = 0.05 * t
        out[name] = t
    return out


def bs256_layer_case():
    """Inputs of the benchmark-batch single-layer fixture (tests/golden/layer_bs256.npz): 256 ragged graphs of up to 64
    atoms at the real widths, cotangents for the three outputs, and the sampling pattern the fixture stores (rows of the
    outputs / entries of every gradient).  Drawn from fixed generators so that the fixture itself only holds outputs."""
    g = torch.Generator().manual_seed(256)
    bs, n, dx, de, dy = 256, 64, 256, 64, 1024
    sizes = torch.randint(30, n + 1, (bs,), generator=g)
    sizes[0] = n
    mask = torch.arange(n)[None, :] < sizes[:, None]
    X = torch.randn(bs, n, dx, generator=g) * mask.unsqueeze(-1)
    E = torch.randn(bs, n, n, de, generator=g)
    E = (E + E.transpose(1, 2)) / 2 * (mask.unsqueeze(1) & mask.unsqueeze(2)).unsqueeze(-1)
    y = torch.randn(bs, dy, generator=g)
    cX = torch.randn(bs, n, dx, generator=g)
    cE = torch.randn(bs, n, n, de, generator=g)
    cy = torch.randn(bs, dy, generator=g)
    pick = {"rows_X": torch.randperm(bs * n, generator=g)[:1024], "rows_E": torch.randperm(bs * n * n, generator=g)[:4096],
            "rows_y": torch.randperm(bs, generator=g)[:32]}
    return dict(X=X, E=E, y=y, mask=mask, cX=cX, cE=cE, cy=cy, pick=pick)


def sample_entries(name, numel, k=2048):
    """Indices of the (at most k) entries of a gradient tensor that the benchmark-batch fixture stores."""
    g = torch.Generator().manual_seed(zlib.crc32(("pick:" + name).encode()))
    return torch.randperm(numel, generator=g)[:k] if numel > k else torch.arange(numel)
