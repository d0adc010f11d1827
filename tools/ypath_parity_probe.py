"""Linear layers of the global-feature path at the benchmark batch (M = 256 rows) against float64 torch: forward, input and
weight gradients, max error relative to each tensor's largest magnitude.  The ReLU gates of the float64 run are taken from
the kernel's own output (at 0.5 M outputs one pre-activation in a few runs lies within fp32 rounding of zero).  GPU box only."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops as O
g = torch.Generator().manual_seed(0)
for M in (64, 160, 256):
    for K, N in ((1024, 2048), (1024, 1024), (2048, 1024), (256, 1024), (1024, 256), (1037, 2048), (2048, 1037)):
        x = torch.randn(M, K, generator=g).cuda().requires_grad_(True)
        w = (torch.randn(N, K, generator=g) / K ** 0.5).cuda().requires_grad_(True)
        b = torch.randn(N, generator=g).cuda().requires_grad_(True)
        dy = torch.randn(M, N, generator=g).cuda()
        for act in (0, 1):
            x.grad = w.grad = b.grad = None
            yv = O.linear(x, w, b, act=act)
            yv.backward(dy)
            xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
            yr = torch.nn.functional.linear(xd, wd, bd)
            if act:                                    # gates as the kernel saw them: a pre-activation within rounding of zero
                yr = yr * (yv.detach() > 0)             # is a different FUNCTION on the other side, not an error
            yr.backward(dy.double())
            rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
            errs = (rel(yv, yr), rel(x.grad, xd.grad), rel(w.grad, wd.grad), rel(b.grad, bd.grad))
            flag = "  <-- " if max(errs) > 1e-5 else ""
            print(f"M={M:4d} K={K:5d} N={N:5d} act={act}: fwd {errs[0]:.1e} dx {errs[1]:.1e} dw {errs[2]:.1e} db {errs[3]:.1e}{flag}", flush=True)
