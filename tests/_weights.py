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
