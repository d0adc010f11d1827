"""Times the fused edge block (projection + attention) forward and backward at the cfg3 shape (GPU box only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "mb-vae-dti_amd")); sys.path.insert(0, ROOT)
import torch
from mbvae_amd import ops

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = "cuda"
bs, n, dx, de = 256, 64, 256, 64
mask = torch.ones(bs, n, dtype=torch.bool, device=dev)
E = torch.randn(bs, n, n, de, device=dev, requires_grad=True)
w = (torch.randn(2 * dx, de, device=dev) / 8).requires_grad_(True); b = torch.zeros(2 * dx, device=dev, requires_grad=True)
QKV = torch.randn(bs, n, 3 * dx, device=dev, requires_grad=True)
ya = torch.randn(bs, dx, device=dev, requires_grad=True); ym = torch.randn(bs, dx, device=dev, requires_grad=True)
dP, dW = torch.randn(bs, n, n, dx, device=dev), torch.randn(bs, n, dx, device=dev)
tf = tb = 0.0
for it in range(iters + 2):
    e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    e[0].record()
    P, wV, st, al = ops.edge_block(E, w, b, QKV, ya, ym, mask, 8)
    e[1].record()
    torch.autograd.backward((P, wV), (dP, dW))
    e[2].record()
    torch.cuda.synchronize()
    if it >= 2:
        tf += e[0].elapsed_time(e[1]); tb += e[1].elapsed_time(e[2])
    for t in (E, w, b, QKV, ya, ym):
        t.grad = None
    del P, wV, st, al
print(f"edge block fwd {1e3 * tf / iters:8.1f} us   bwd (attention + dgrad + wgrad) {1e3 * tb / iters:8.1f} us")
