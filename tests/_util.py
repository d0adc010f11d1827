"""Helpers shared by the tests: fixture loading and the parity comparison."""
import os

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def load_golden(name):
    """Fixture file -> dict of torch tensors; keys 'a/b' become nested dicts {'a': {'b': ...}}."""
    z = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
    out = {}
    for k in z.files:
        v = z[k]
        t = torch.from_numpy(v) if v.dtype.kind in "fiub" else v
        if "/" in k:
            a, b = k.split("/", 1)
            out.setdefault(a, {})[b] = t
        else:
            out[k] = t
    return out


def assert_close(a, b, rtol=1e-5, atol=None, what=""):
    """max |a - b| <= rtol * max |b|: the error relative to the reference tensor's own scale (1e-5 unless a test states
    and justifies another figure).  No floor on the scale: a gradient tensor of magnitude 1e-3 must be right to 1e-8.
    atol (absolute) replaces the relative bound where the reference is exactly zero or a difference of O(1) terms."""
    a = a.detach().double().cpu()
    b = b.detach().double().cpu()
    assert a.shape == b.shape, f"{what}: shape {tuple(a.shape)} vs {tuple(b.shape)}"
    assert bool(torch.isfinite(a).all()), f"{what}: non-finite values"
    scale = float(b.abs().max()) if b.numel() else 0.0
    err = float((a - b).abs().max()) if b.numel() else 0.0
    cap = os.environ.get("MBVAE_TEST_RTOL_CAP")          # survey aid: which checks need more than a given bound
    if cap:
        rtol = min(rtol, float(cap))
    tol = rtol * scale if atol is None else atol
    report = os.environ.get("MBVAE_TEST_REPORT")         # survey aid: one line per check with its relative error
    if report:
        with open(report, "a") as f:
            f.write(f"{err / max(scale, 1e-300):.3e}\t{rtol:.1e}\t{'abs' if atol is not None else 'rel'}\t"
                    f"{os.environ.get('PYTEST_CURRENT_TEST', '').split(' ')[0]}\t{what}\n")
    assert err <= tol, f"{what}: max abs err {err:.3e} > {tol:.3e} (scale {scale:.3e}, rel {err / max(scale, 1e-300):.2e})"
